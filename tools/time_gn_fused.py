"""Fused GroupNorm-apply+SiLU in the halo conv's staging (XT = 1 / 2) against the separate pass + conv, per layer, by hipEvents.
   python tools/time_gn_fused.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_tagger_amd import _lib
ctx = _lib.Context(0); dev = torch.device("cuda:0")
vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
def timed(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (B, H, W, C) in ((8, 1024, 1024, 128), (8, 512, 512, 256), (16, 256, 256, 512)):
    torch.manual_seed(0)
    x32 = torch.randn(B, H, W, C, device=dev)
    x16 = x32.to(torch.bfloat16)
    w = (torch.randn(C, 3, 3, C, device=dev) * (C * 9) ** -0.5).to(torch.bfloat16)
    b = torch.zeros(C, device=dev)
    ss = torch.stack([torch.ones(B, C), torch.zeros(B, C)], -1).to(dev).contiguous()
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    o16 = torch.empty(B, H, W, C, device=dev, dtype=torch.bfloat16)
    act = torch.empty(B, H, W, C, device=dev, dtype=torch.bfloat16)
    ws = torch.empty(ctx.lib.vt_op_groupnorm_workspace_bytes(B, H * W, C) + 256, dtype=torch.uint8, device=dev)
    gn = lambda: ctx.call("vt_op_groupnorm", vp(x16), _lib.VT_BF16, B, H * W, C, 32, 1e-6, vp(gam), vp(bet), 1, vp(act), vp(ws), None)
    cv = lambda: ctx.call("vt_op_conv2d", vp(act), vp(w), vp(b), None, None, vp(o16), B, H, W, C, C, 3, 1, 1, 1, None)
    f32 = lambda: ctx.call("vt_op_norm_silu_conv3x3", vp(x32), _lib.VT_F32, vp(ss), vp(w), vp(b), None, None, vp(o16), B, H, W, C, C, None)
    f16 = lambda: ctx.call("vt_op_norm_silu_conv3x3", vp(x16), _lib.VT_BF16, vp(ss), vp(w), vp(b), None, None, vp(o16), B, H, W, C, C, None)
    t_gn, t_cv, t_f32, t_f16 = timed(gn), timed(cv), timed(f32), timed(f16)
    print(f"B{B} {H}x{W} C{C}: groupnorm op (stats + apply) {t_gn:.3f} ms + conv {t_cv:.3f} ms = {t_gn + t_cv:.3f} | fused fp32-in {t_f32:.3f} | fused bf16-in {t_f16:.3f}", flush=True)
