"""Workload for SQ counter passes on the question "why is GroupNorm-apply+SiLU fused into the halo conv's staging slower
than the separate HBM-bound pass?" (DESIGN.md section 4.1): the same two layers in three forms.
   rocprofv3 --pmc <counters> --kernel-trace -- python3 tools/pmc_gn_fused.py raw|f32norm|bf16norm
raw      = conv3x3_halo_kernel<2,2,0,8,4> on a bf16 activation (what ships; the separate gn_apply pass runs first)
f32norm  = conv3x3_halo_kernel<.,.,1,8,6>: fp32 input normalised + SiLU'ed in the staging (XT = 1)
bf16norm = conv3x3_halo_kernel<.,.,2,8,6>: bf16 input normalised + SiLU'ed in the staging (XT = 2)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_tagger_amd import _lib
mode = sys.argv[1] if len(sys.argv) > 1 else "raw"
ctx = _lib.Context(0); dev = torch.device("cuda:0")
vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
for (B, H, W, C) in ((8, 1024, 1024, 128), (8, 512, 512, 256)):
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
    for _ in range(4):
        if mode == "raw":
            ctx.call("vt_op_groupnorm", vp(x16), _lib.VT_BF16, B, H * W, C, 32, 1e-6, vp(gam), vp(bet), 1, vp(act), vp(ws), None)
            ctx.call("vt_op_conv2d", vp(act), vp(w), vp(b), None, None, vp(o16), B, H, W, C, C, 3, 1, 1, 1, None)
        elif mode == "f32norm":
            ctx.call("vt_op_norm_silu_conv3x3", vp(x32), _lib.VT_F32, vp(ss), vp(w), vp(b), None, None, vp(o16), B, H, W, C, C, None)
        else:
            ctx.call("vt_op_norm_silu_conv3x3", vp(x16), _lib.VT_BF16, vp(ss), vp(w), vp(b), None, None, vp(o16), B, H, W, C, C, None)
    torch.cuda.synchronize()
