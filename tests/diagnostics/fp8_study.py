"""Numerics study for BASELINE.json configs[4] (fp8 operands on the CDNA4 fp8 MFMA path).  CPU only; test infrastructure.

How far do the latents AND the logits (the quantity north_star's fp8 target line constrains: "logits within 1e-2 of CPU
reference") move when the 3x3 convolutions of the encoder take fp8 e4m3 operands instead of bf16?

Quantisers (what the hardware can consume):
  weights      per-output-channel absmax scale, e4m3 (the scale multiplies the accumulator in the epilogue);
  activations  'tensor'  one power-of-two scale per layer (static: GroupNorm + SiLU outputs are bounded), e4m3;
               'mx'      one e8m0 (power-of-two) scale per 32 consecutive channels of a pixel -- the block format
                         v_mfma_scale_f32_16x16x128_f8f6f4 consumes;
               'rowmax'  per-pixel absmax scale (upper bound of what finer scaling could buy).
The attention projections / QK^T / PV and conv_in stay bf16 / fp32 in every variant (12 % of the FLOPs).

  python tests/diagnostics/fp8_study.py [res=256] [per_layer=1]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F

from oracle import decoder_ref, encoder_ref
from vae_tagger_amd import synth

E4M3_MAX = 448.0


def e4m3(t):
    return t.clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).to(torch.float32)


def q_weight(w):
    s = w.abs().amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-12) / E4M3_MAX
    return e4m3(w / s) * s


def q_act(x, mode):
    if mode == "tensor":
        s = 2.0 ** torch.ceil(torch.log2(x.abs().amax().clamp_min(1e-12) / E4M3_MAX))
        return e4m3(x / s) * s
    if mode == "rowmax":
        s = x.abs().amax(dim=1, keepdim=True).clamp_min(1e-12) / E4M3_MAX
        return e4m3(x / s) * s
    if mode == "mx":                                    # e8m0 scale per 32 channels of a pixel (NHWC k-blocks)
        b, c, h, w = x.shape
        xb = x.reshape(b, c // 32, 32, h, w)
        s = 2.0 ** torch.ceil(torch.log2(xb.abs().amax(dim=2, keepdim=True).clamp_min(2.0 ** -100) / E4M3_MAX))
        return (e4m3(xb / s) * s).reshape(b, c, h, w)
    raise ValueError(mode)


class Policy:
    """Which conv layers run on fp8 operands, and how activations are scaled."""

    def __init__(self, layers=(), act="tensor"):
        self.layers, self.act = set(layers), act


POLICY = Policy()
_orig_conv = encoder_ref._conv


def _conv(x, sd, name, q, stride=1, padding=1):
    w = sd[name + ".weight"]
    if name in POLICY.layers and w.shape[-1] == 3:
        return F.conv2d(q_act(x, POLICY.act), q_weight(w), sd[name + ".bias"], stride=stride, padding=padding)
    return _orig_conv(x, sd, name, q, stride, padding)


encoder_ref._conv = _conv

ATTN_FP8 = {"p": False, "v": False, "qk": False}
_orig_attention = encoder_ref._attention


def _attention(h, sd, p, q):
    """encoder_ref._attention with optional e4m3 P (un-normalised numerators exp(s - rowmax), scaled by 64), V and q / k."""
    import math
    if not any(ATTN_FP8.values()):
        return _orig_attention(h, sd, p, q)
    b, c, hh, ww = h.shape
    x = F.group_norm(h, encoder_ref.GN_GROUPS, sd[p + ".group_norm.weight"], sd[p + ".group_norm.bias"], encoder_ref.GN_EPS)
    x = q(x).reshape(b, c, hh * ww).transpose(1, 2)
    qq = q(F.linear(x, q(sd[p + ".to_q.weight"]), sd[p + ".to_q.bias"]))
    kk = q(F.linear(x, q(sd[p + ".to_k.weight"]), sd[p + ".to_k.bias"]))
    vv = q(F.linear(x, q(sd[p + ".to_v.weight"]), sd[p + ".to_v.bias"]))
    if ATTN_FP8["qk"]:
        qq, kk = q_act(qq, "tensor"), q_act(kk, "tensor")
    scores = torch.matmul(qq, kk.transpose(1, 2)) * (1.0 / math.sqrt(c))
    num = torch.exp(scores - scores.amax(dim=-1, keepdim=True))
    den = num.sum(dim=-1, keepdim=True)                       # the fp32 numerators' sum, as the Q.K^T epilogue takes it
    if ATTN_FP8["p"]:
        num = e4m3(num * 64.0) / 64.0
    else:
        num = q(num)
    if ATTN_FP8["v"]:
        vv = q_act(vv, "tensor")
    o = q(torch.matmul(num, vv) / den)
    o = F.linear(o, q(sd[p + ".to_out.0.weight"]), sd[p + ".to_out.0.bias"])
    return o.transpose(1, 2).reshape(b, c, hh, ww) + h


encoder_ref._attention = _attention


def conv_layers(sd):
    """3x3 convs of the encoder except conv_in / conv_out, with (cin, cout)."""
    out = []
    for k, v in sd.items():
        if k.endswith(".weight") and v.dim() == 4 and v.shape[-1] == 3 and "conv_in" not in k and "conv_out" not in k:
            out.append((k[:-7], v.shape[1], v.shape[0]))
    return out


def main():
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    per_layer = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    n_tags = 10000
    sd = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
    sd_d = synth.synth_state_dict(synth.attention_decoder_manifest(n_tags), seed=1)
    x = synth.synth_images(1, res, res, seed=3)
    torch.set_grad_enabled(False)

    def run(policy):
        global POLICY
        POLICY = policy
        lat = encoder_ref.vae_wrapper_encode(sd, x, emulate_bf16=True)
        return lat, decoder_ref.attention_decoder_forward(sd_d, lat)

    POLICY_NONE = Policy()
    t0 = time.time()
    global POLICY
    POLICY = POLICY_NONE
    ref_lat = encoder_ref.vae_wrapper_encode(sd, x)                       # fp32 oracle
    ref_lg = decoder_ref.attention_decoder_forward(sd_d, ref_lat)
    print(f"{res}x{res}: latent std {ref_lat.std():.3f} range [{ref_lat.min():.2f}, {ref_lat.max():.2f}]; logits std {ref_lg.std():.3f} "
          f"range [{ref_lg.min():.2f}, {ref_lg.max():.2f}]  ({time.time() - t0:.0f} s per oracle run)", flush=True)

    def report(label, policy):
        lat, lg = run(policy)
        dl, dg = (lat - ref_lat), (lg - ref_lg)
        print(f"{label:58s} |dlatent| max {dl.abs().max():.4f} rms {dl.pow(2).mean().sqrt():.5f}   |dlogit| max {dg.abs().max():.5f} "
              f"rms {dg.pow(2).mean().sqrt():.6f}", flush=True)
        return dl.abs().max().item(), dg.abs().max().item()

    layers = conv_layers(sd)
    allc = [n for n, ci, co in layers]
    big = [n for n, ci, co in layers if ci >= 256]                       # K >= 2304: the 256 / 512-channel layers
    s1 = [n for n, ci, co in layers if "downsamplers" not in n]
    report("bf16 operands everywhere (the shipped path)", Policy())
    for act in ("tensor", "mx", "rowmax"):
        report(f"fp8 e4m3, all 3x3 convs, act scale = {act}", Policy(allc, act))
    report("fp8 e4m3, Cin >= 256 layers only, act scale = tensor", Policy(big, "tensor"))
    report("fp8 e4m3, Cin >= 256 layers only, act scale = mx", Policy(big, "mx"))
    report("fp8 e4m3, stride-1 resnet convs only, act scale = tensor", Policy(s1, "tensor"))
    for label, cfg in (("+ attention P in e4m3", dict(p=True)), ("+ attention P and V in e4m3", dict(p=True, v=True)),
                       ("+ attention P, V, q, k in e4m3", dict(p=True, v=True, qk=True))):
        ATTN_FP8.update(dict(p=False, v=False, qk=False)); ATTN_FP8.update(cfg)
        report("fp8 stride-1 resnet convs " + label, Policy(s1, "tensor"))
    ATTN_FP8.update(dict(p=False, v=False, qk=False))
    if per_layer:
        print("per-layer sensitivity (ONE layer on fp8 operands, act scale = tensor):", flush=True)
        rows = []
        for n, ci, co in layers:
            dl, dg = report(f"  {n} ({ci}->{co})", Policy([n], "tensor"))
            rows.append((dg, dl, n))
        rows.sort(reverse=True)
        print("most sensitive layers by |dlogit|: " + ", ".join(f"{n.replace('encoder.', '')} {dg:.1e}" for dg, dl, n in rows[:6]), flush=True)
        keep = {n for _, _, n in rows[:4]}
        report("fp8 e4m3 everywhere except the 4 most sensitive layers", Policy([n for n in allc if n not in keep], "tensor"))


if __name__ == "__main__":
    main()
