"""One conv layer, few launches (for PMC passes).  python tools/bench_one.py MODE B H W CIN COUT"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_tagger_amd import _lib
mode, B, H, W, Cin, Cout = sys.argv[1], *map(int, sys.argv[2:7])
ctx = _lib.Context(0); dev = torch.device("cuda:0")
vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
g = torch.Generator().manual_seed(0)
x16 = torch.randn(B, H, W, Cin, generator=g).to(dev, torch.bfloat16)
w = (torch.randn(Cout, 3, 3, Cin, generator=g) * (Cin * 9) ** -0.5).to(dev, torch.bfloat16)
b = torch.zeros(Cout, device=dev); res = torch.randn(B, H, W, Cout, device=dev)
o32 = torch.empty(B, H, W, Cout, device=dev); o16 = torch.empty(B, H, W, Cout, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    if mode == "raw": ctx.call("vt_op_conv2d", vp(x16), vp(w), vp(b), None, None, vp(o16), B, H, W, Cin, Cout, 3, 1, 1, 1, None)
    else: ctx.call("vt_op_conv2d", vp(x16), vp(w), vp(b), vp(res), vp(o32), None, B, H, W, Cin, Cout, 3, 1, 1, 1, None)
torch.cuda.synchronize()
