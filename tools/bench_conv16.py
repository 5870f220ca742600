import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_tagger_amd import _lib
ctx = _lib.Context(0); dev = torch.device("cuda:0")
vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
import torch.nn.functional as F
for waves in (8, 16):
    ctx.call("vt_set_flag", 3, waves)
    for (B, H, W, Cin, Cout) in [(8, 1024, 1024, 128, 128), (8, 512, 512, 256, 256), (8, 256, 256, 512, 512)]:
        g = torch.Generator().manual_seed(0)
        x16 = torch.randn(B, H, W, Cin, generator=g).to(dev, torch.bfloat16)
        w = (torch.randn(Cout, 3, 3, Cin, generator=g) * (Cin * 9) ** -0.5).to(dev, torch.bfloat16)
        b = torch.zeros(Cout, device=dev); o16 = torch.empty(B, H, W, Cout, device=dev, dtype=torch.bfloat16)
        call = lambda: ctx.call("vt_op_conv2d", vp(x16), vp(w), vp(b), None, None, vp(o16), B, H, W, Cin, Cout, 3, 1, 1, 1, None)
        for _ in range(3): call()
        torch.cuda.synchronize()
        if H == 256:   # correctness spot check against torch on a crop
            ref = F.conv2d(x16[:1, :40, :40].permute(0, 3, 1, 2).float().cpu(), w.permute(0, 3, 1, 2).float().cpu(), padding=1)
            got = o16[:1, :38, :38].permute(0, 3, 1, 2).float().cpu()
            print("   max err vs torch (crop):", (got - ref[:, :, :38, :38]).abs().max().item())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): call()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"waves {waves:2d} B{B} {H}x{W} {Cin}->{Cout}: {ms:7.3f} ms {2.0*B*H*W*Cout*9*Cin/ms/1e9:7.1f} TFLOP/s", flush=True)
ctx.call("vt_set_flag", 3, 8)
