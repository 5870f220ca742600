"""Diagnostic: phase cycle shares of the halo conv K-step (needs the EXP_STAMP build).  python tools/stamp_conv.py lib.so"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_tagger_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
ctx = _lib.Context(0); dev = torch.device("cuda:0")
vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
for (B, H, W, Cin, Cout) in [(8, 256, 256, 512, 512), (8, 1024, 1024, 128, 128)]:
    g = torch.Generator().manual_seed(0)
    x16 = torch.randn(B, H, W, Cin, generator=g).to(dev, torch.bfloat16)
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) * (Cin * 9) ** -0.5).to(dev, torch.bfloat16)
    b = torch.zeros(Cout, device=dev); o16 = torch.empty(B, H, W, Cout, device=dev, dtype=torch.bfloat16)
    rows = 16 if Cout % 256 == 0 else 32
    nblk = ((W + 15) // 16) * ((H + rows - 1) // rows) * (Cout // (256 if Cout % 256 == 0 else 128)) * B
    dbg = torch.zeros(nblk * 8 * 8, dtype=torch.int64, device=dev)
    ctx.lib.vt_set_debug_buffer(vp(dbg))
    for _ in range(3):
        ctx.call("vt_op_conv2d", vp(x16), vp(w), vp(b), None, None, vp(o16), B, H, W, Cin, Cout, 3, 1, 1, 1, None)
    torch.cuda.synchronize()
    d = dbg.view(nblk, 8, 8).double().cpu()
    steps = d[:, :, 4].mean().item()
    names = ["vmcnt wait", "barrier", "issue DMA", "MFMA+refill"]
    per = [d[:, :, i].sum().item() / d[:, :, 4].sum().item() for i in range(4)]
    tot = sum(per)
    print(f"B{B} {H}x{W} {Cin}->{Cout}: steps/wave {steps:.0f}; cycles/step " + ", ".join(f"{n} {p:.0f}" for n, p in zip(names, per))
          + f", total {tot:.0f} (MFMA-only bound: 1024 per SIMD pair); mainloop {d[:, :, 5].mean().item():.0f} cyc, epilogue {d[:, :, 6].mean().item():.0f} cyc")
    for wv in (0, 4):
        pw = [d[:, wv, i].sum().item() / d[:, wv, 4].sum().item() for i in range(4)]
        print(f"   wave {wv}: " + ", ".join(f"{n} {p:.0f}" for n, p in zip(names, pw)))
ctx.lib.vt_set_debug_buffer(None)
