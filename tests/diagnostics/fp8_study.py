"""Numerics study for BASELINE.json configs[4] (fp8 weights on the fp8 MFMA path): how far do the latents move when the
MFMA operands are rounded to fp8 e4m3 instead of bf16?  CPU only, small image; uses the oracle's rounding hooks.
   python tests/diagnostics/fp8_study.py            (prints max |d latent| vs the fp32 oracle for bf16, fp8 weights, fp8 weights+activations)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import encoder_ref
from vae_tagger_amd import synth

def fp8(t, per_row=False):
    """round to e4m3 with a power-of-two-free absmax scale (per output channel for weights, per tensor otherwise)"""
    if per_row and t.dim() >= 2:
        s = t.abs().amax(dim=tuple(range(1, t.dim())), keepdim=True).clamp_min(1e-12) / 448.0
    else:
        s = t.abs().amax().clamp_min(1e-12) / 448.0
    return (t / s).to(torch.float8_e4m3fn).to(torch.float32) * s

class QW(encoder_ref._Q):
    def __init__(self, mode): self.mode = mode; self.on = True
    def __call__(self, t):
        is_weight = t.dim() in (2, 4) and t.requires_grad is False and getattr(t, "_is_w", False)
        if self.mode == "w8":      # fp8 weights, bf16 activations
            return fp8(t, True) if is_weight else encoder_ref._bf16(t)
        return fp8(t, True) if is_weight else fp8(t)

sd = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
for k, v in sd.items():
    if k.endswith(".weight") and v.dim() in (2, 4): v._is_w = True
x = synth.synth_images(1, 128, 128, seed=3)
ref = encoder_ref.vae_wrapper_encode(sd, x)
print(f"latent range [{ref.min().item():.3f}, {ref.max().item():.3f}], std {ref.std().item():.3f}")
print(f"bf16 operands (the shipped path):        max |d latent| = {(encoder_ref.vae_wrapper_encode(sd, x, emulate_bf16=True) - ref).abs().max().item():.4f}")
orig = encoder_ref._Q
for mode, label in (("w8", "fp8 e4m3 weights, bf16 activations:   "), ("w8a8", "fp8 e4m3 weights AND activations:     ")):
    encoder_ref._Q = lambda on, m=mode: QW(m)
    sdq = {k: v for k, v in sd.items()}
    got = encoder_ref.vae_wrapper_encode(sdq, x, emulate_bf16=True)
    print(f"{label} max |d latent| = {(got - ref).abs().max().item():.4f}")
encoder_ref._Q = orig
