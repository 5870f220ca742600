"""The three Downsample2D convs of the encoder at batch 16 (stride 2, pad (0,1,0,1)) on the phase-plane halo kernel and on the
generic implicit GEMM (vt_set_flag 13):   python tools/bench_s2.py [lib.so]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_tagger_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
ctx = _lib.Context(0); dev = torch.device("cuda:0")
vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
print("lib:", _lib.LIB_PATH, flush=True)
ns = ctx.lib.vt_profile_num_configs()
for (B, H, W, C) in ((16, 1024, 1024, 128), (16, 512, 512, 256), (16, 256, 256, 512)):
    torch.manual_seed(0)
    x = torch.randn(B, H, W, C, device=dev).to(torch.bfloat16)
    w = (torch.randn(C, 3, 3, C, device=dev) * (C * 9) ** -0.5).to(torch.bfloat16)
    b = torch.zeros(C, device=dev)
    o16 = torch.empty(B, H // 2, W // 2, C, device=dev, dtype=torch.bfloat16)
    for flag in (1, 0):
        ctx.call("vt_set_flag", 13, flag)
        call = lambda: ctx.call("vt_op_conv2d", vp(x), vp(w), vp(b), None, None, vp(o16), B, H, W, C, C, 3, 2, 0, 1, None)
        for _ in range(3): call()
        torch.cuda.synchronize()
        la = (ctypes.c_longlong * ns)(); ms = (ctypes.c_double * ns)(); fl = (ctypes.c_double * ns)(); nm = (ctypes.c_char_p * ns)()
        ctx.call("vt_profile_begin")
        for _ in range(10): call()
        ctx.call("vt_profile_end", ns, la, ms, fl, nm)
        for k in range(ns - 1):
            if la[k]:
                print(f"{C}->{C} @{H}x{W} s2  {'phase-plane' if flag else 'generic    '}  {nm[k].decode():36s} {ms[k] / la[k]:7.3f} ms  {fl[k] / ms[k] / 1e9:7.1f} TF/s", flush=True)
ctx.call("vt_set_flag", 13, 1)
