"""ORACLE (test infrastructure, not product code): CPU fp32 restatement of the FLUX
AutoencoderKL *encoder* forward that the reference calls at
diffusers_vae_loader.py:79 (`self.vae.encode(x).latent_dist`) and post-scales at :80-84.

PARITY UNPINNED for the third-party half: the arithmetic lives in `diffusers`
(requirements.txt:3 `diffusers>=0.21.0`, effective ~0.30 per the config stamp at
diffusers_vae_loader.py:105), which is not installed, vendored or fetchable here, and
the reference holds no golden vectors for it.  This file restates diffusers' published
AutoencoderKL encoder topology (Encoder -> DownEncoderBlock2D x4 -> UNetMidBlock2D ->
conv_norm_out/SiLU/conv_out -> DiagonalGaussianDistribution.mode) with the
hyper-parameters the reference passes (diffusers_vae_loader.py:8-35, :102-134).  Each
primitive is torch's own CPU kernel; the wiring is what cannot be checked offline.  What
pins it: parameter count 34,274,208 and the key/shape manifest (tests/test_oracle.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import math

import torch
import torch.nn.functional as F

GN_GROUPS = 32          # norm_num_groups, diffusers_vae_loader.py:27
GN_EPS = 1e-6           # diffusers Encoder/ResnetBlock2D eps for VAE blocks
SCALING_FACTOR = 0.3611  # diffusers_vae_loader.py:29
SHIFT_FACTOR = 0.1159    # diffusers_vae_loader.py:30


def _bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)


class _Q:
    """Rounding policy: identity (fp32 oracle) or bf16 at the points where the HIP path feeds bf16
    to the matrix cores (weights, GroupNorm outputs, conv1 outputs, q/k/v/P/o); the residual stream
    stays fp32 there too.  Numerics study only -- never the parity target."""

    def __init__(self, emulate_bf16):
        self.on = emulate_bf16

    def __call__(self, t):
        return _bf16(t) if self.on else t


def _gn(x, sd, name, q, silu):
    y = F.group_norm(x, GN_GROUPS, sd[name + ".weight"], sd[name + ".bias"], GN_EPS)
    if silu:
        y = F.silu(y)
    return q(y)


def _conv(x, sd, name, q, stride=1, padding=1):
    return F.conv2d(x, q(sd[name + ".weight"]), sd[name + ".bias"], stride=stride, padding=padding)


def _resnet(h, sd, p, q):
    # diffusers ResnetBlock2D, no time embedding, output_scale_factor 1:
    # norm1 -> silu -> conv1 -> norm2 -> silu -> (dropout p=0) -> conv2 ; + shortcut(x)
    t = _gn(h, sd, p + ".norm1", q, True)
    t = q(_conv(t, sd, p + ".conv1", q))
    t = _gn(t, sd, p + ".norm2", q, True)
    t = _conv(t, sd, p + ".conv2", q)
    if (p + ".conv_shortcut.weight") in sd:
        s = _conv(q(h), sd, p + ".conv_shortcut", q, padding=0)
    else:
        s = h
    return t + s


def _attention(h, sd, p, q):
    # diffusers Attention as built by UNetMidBlock2D for a VAE: heads = 1, dim_head = C,
    # group_norm(32, eps 1e-6), residual_connection=True, rescale_output_factor=1, bias=True.
    b, c, hh, ww = h.shape
    x = F.group_norm(h, GN_GROUPS, sd[p + ".group_norm.weight"], sd[p + ".group_norm.bias"], GN_EPS)
    x = q(x).reshape(b, c, hh * ww).transpose(1, 2)          # [B, S, C]
    qq = q(F.linear(x, q(sd[p + ".to_q.weight"]), sd[p + ".to_q.bias"]))
    kk = q(F.linear(x, q(sd[p + ".to_k.weight"]), sd[p + ".to_k.bias"]))
    vv = q(F.linear(x, q(sd[p + ".to_v.weight"]), sd[p + ".to_v.bias"]))
    scores = torch.matmul(qq, kk.transpose(1, 2)) * (1.0 / math.sqrt(c))
    probs = torch.softmax(scores, dim=-1)
    o = q(torch.matmul(q(probs), vv))
    o = F.linear(o, q(sd[p + ".to_out.0.weight"]), sd[p + ".to_out.0.bias"])
    o = o.transpose(1, 2).reshape(b, c, hh, ww)
    return o + h


def encoder_moments(sd, x, emulate_bf16=False, n_down=4, layers_per_block=2, taps=None):
    """x fp32 [B,3,H,W] -> moments fp32 [B, 2*latent, H/8, W/8] (mean | logvar).
    `taps`, if a dict, receives named intermediate activations."""
    q = _Q(emulate_bf16)
    sd = {k: v.to(torch.float32) for k, v in sd.items() if k.startswith("encoder.")}
    h = F.conv2d(x, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)   # fp32 on the HIP path too
    if taps is not None:
        taps["conv_in"] = h
    for i in range(n_down):
        for j in range(layers_per_block):
            h = _resnet(h, sd, f"encoder.down_blocks.{i}.resnets.{j}", q)
        if taps is not None:
            taps[f"down{i}"] = h
        d = f"encoder.down_blocks.{i}.downsamplers.0.conv"
        if (d + ".weight") in sd:
            # diffusers Downsample2D(padding=0): F.pad(x, (0,1,0,1)) then conv stride 2
            h = _conv(F.pad(q(h), (0, 1, 0, 1)), sd, d, q, stride=2, padding=0)
    h = _resnet(h, sd, "encoder.mid_block.resnets.0", q)
    h = _attention(h, sd, "encoder.mid_block.attentions.0", q)
    if taps is not None:
        taps["mid_attn"] = h
    h = _resnet(h, sd, "encoder.mid_block.resnets.1", q)
    h = _gn(h, sd, "encoder.conv_norm_out", q, True)
    return _conv(h, sd, "encoder.conv_out", q)


def vae_wrapper_encode(sd, x, emulate_bf16=False, taps=None):
    """DiffusersVAEWrapper.encode (diffusers_vae_loader.py:78-86):
    latent_dist.mode() * scaling_factor + shift_factor  (mode = mean = first half of channels)."""
    moments = encoder_moments(sd, x, emulate_bf16=emulate_bf16, taps=taps)
    mean = moments[:, : moments.shape[1] // 2]
    return mean * SCALING_FACTOR + SHIFT_FACTOR


def encoder_flops(height, width):
    """Algorithmic FLOPs (2*MAC) of one encoder forward, SURVEY.md section 8(d)."""
    co = (128, 256, 512, 512)
    f = 2 * height * width * 27 * 128
    hh, ww, ci = height, width, 128
    for i, c in enumerate(co):
        f += 2 * hh * ww * 9 * ci * c + 2 * hh * ww * 9 * c * c          # resnet 0
        if ci != c:
            f += 2 * hh * ww * ci * c
        f += 2 * (2 * hh * ww * 9 * c * c)                                 # resnet 1
        ci = c
        if i != 3:
            hh, ww = hh // 2, ww // 2
            f += 2 * hh * ww * 9 * c * c
    s = hh * ww
    f += 4 * (2 * s * 9 * 512 * 512)                                       # mid resnets
    f += 4 * (2 * s * 512 * 512) + 2 * (2 * s * s * 512)                   # attention
    f += 2 * s * 9 * 512 * 32
    return f
