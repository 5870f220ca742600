"""ORACLE (test infrastructure, not product code): CPU fp32 restatement of the reference's
classification decoders, written functionally over a plain state-dict.

Pinned: oracle/make_goldens.py imports the reference's own modules.py in the build
container, runs it on seeded weights/inputs, asserts this restatement agrees to 1e-5 and
commits the reference's outputs under tests/golden/decoder_*.npz.  tests/test_oracle.py
re-checks this file against those fixtures everywhere (the reference never travels).

Follows (reference modules.py):
  SpatialAttention.forward                 :36-47
  feature_compress                         :377-382
  MultiHeadSelfAttention.forward           :66-91
  CrossAttention.forward                   :107-124 (used at :451-459)
  AttentionClassificationDecoder.forward   :424-468
  ClassificationDecoder.forward            :333-349
  get_confidence                           :470-475 / :351-356

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-5   # nn.BatchNorm2d default
LN_EPS = 1e-5   # nn.LayerNorm default


def spatial_attention(sd, x, p="spatial_attention."):
    w0 = sd[p + "channel_att.0.weight"]
    w2 = sd[p + "channel_att.2.weight"]

    def mlp(v):
        return F.conv2d(F.relu(F.conv2d(v, w0)), w2)

    avg = x.mean(dim=(2, 3), keepdim=True)
    mx = x.amax(dim=(2, 3), keepdim=True)
    x = x * torch.sigmoid(mlp(avg) + mlp(mx))
    sp = torch.cat([x.mean(dim=1, keepdim=True), x.amax(dim=1, keepdim=True)], dim=1)
    gate = torch.sigmoid(F.conv2d(sp, sd[p + "spatial_att.0.weight"], padding=3))
    return x * gate


def feature_compress(sd, x, p="feature_compress."):
    y = F.conv2d(x, sd[p + "0.weight"], sd[p + "0.bias"], padding=1)
    y = F.batch_norm(y, sd[p + "1.running_mean"], sd[p + "1.running_var"],
                     sd[p + "1.weight"], sd[p + "1.bias"], training=False, eps=BN_EPS)
    return F.adaptive_avg_pool2d(F.relu(y), (8, 8))


def self_attention(sd, x, num_heads=8, p="self_attention_post."):
    b, c, h, w = x.shape
    s, hd = h * w, c // num_heads
    xf = x.reshape(b, c, s).transpose(1, 2)
    res = xf                                                  # un-normed residual (:73, :87)
    xn = F.layer_norm(xf, (c,), sd[p + "norm.weight"], sd[p + "norm.bias"], LN_EPS)

    def proj(n):
        y = F.linear(xn, sd[p + n + ".weight"], sd[p + n + ".bias"])
        return y.reshape(b, s, num_heads, hd).transpose(1, 2)

    q, k, v = proj("q_proj"), proj("k_proj"), proj("v_proj")
    att = torch.softmax(torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(hd), dim=-1)
    o = torch.matmul(att, v).transpose(1, 2).reshape(b, s, c)
    o = F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"]) + res
    return o.transpose(1, 2).reshape(b, c, h, w)


def cross_attention(sd, query, key_value, num_heads=8, p="cross_attention."):
    b = query.shape[0]
    e = sd[p + "q_proj.weight"].shape[0]
    hd = e // num_heads
    q = F.linear(query, sd[p + "q_proj.weight"], sd[p + "q_proj.bias"]).reshape(b, 1, num_heads, hd).transpose(1, 2)
    k = F.linear(key_value, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"]).reshape(b, -1, num_heads, hd).transpose(1, 2)
    v = F.linear(key_value, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"]).reshape(b, -1, num_heads, hd).transpose(1, 2)
    att = torch.softmax(torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(hd), dim=-1)
    o = torch.matmul(att, v).transpose(1, 2).reshape(b, e)
    return F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"]) + query


def _mlp(sd, x, n_hidden, act):
    for i in range(n_hidden):
        x = F.linear(x, sd[f"classifier.{4 * i}.weight"], sd[f"classifier.{4 * i}.bias"])
        n = x.shape[-1]
        x = act(F.layer_norm(x, (n,), sd[f"classifier.{4 * i + 1}.weight"],
                             sd[f"classifier.{4 * i + 1}.bias"], LN_EPS))
    j = 4 * n_hidden
    return F.linear(x, sd[f"classifier.{j}.weight"], sd[f"classifier.{j}.bias"])


def attention_decoder_forward(sd, latent, num_heads=8, taps=None):
    """AttentionClassificationDecoder.forward in eval mode (dropout = identity)."""
    x = latent.to(torch.float32)
    if "spatial_attention.channel_att.0.weight" in sd:
        x = spatial_attention(sd, x)
        if taps is not None:
            taps["spatial"] = x
    x = feature_compress(sd, x)
    if taps is not None:
        taps["compress"] = x
    if "self_attention_post.q_proj.weight" in sd:
        x = self_attention(sd, x, num_heads)
        if taps is not None:
            taps["self_attn"] = x
    b = x.shape[0]
    flat = x.reshape(b, -1)
    if "cross_attention.q_proj.weight" in sd:
        query = F.linear(flat, sd["query_generator.weight"], sd["query_generator.bias"])
        kv = x.reshape(b, x.shape[1], -1).transpose(1, 2)
        attended = cross_attention(sd, query, kv, num_heads)
        flat = flat + attended.mean(dim=1, keepdim=True).expand_as(flat)
    return _mlp(sd, flat, 3, F.relu)


def plain_decoder_forward(sd, latent):
    """ClassificationDecoder.forward, use_adaptive_pooling=True (infer_full.py:52-58)."""
    x = F.adaptive_avg_pool2d(latent.to(torch.float32), (4, 4))
    return _mlp(sd, x.reshape(x.shape[0], -1), 2, lambda t: F.leaky_relu(t, 0.2))


def get_confidence(logits):
    """sigmoid then descending sort (modules.py:470-475).  torch.sort there is not stable;
    here ties are broken by ascending tag index so the result is defined."""
    conf = torch.sigmoid(logits)
    # sort on logits (monotone in conf, no fp32 saturation), stable => ascending index on ties
    idx = torch.argsort(logits, dim=-1, descending=True, stable=True)
    return torch.gather(conf, -1, idx), idx
