"""VERDICT round 3, item 8 -- the microbenchmark that decides go / no-go: the bf16 halo conv as ONE wave per SIMD.
`vt_set_flag(ctx, 3, 4)` selects conv3x3_halo_kernel<2,2,0,16,6>: 4 waves per workgroup, one workgroup per CU, each wave a 16-row x 16-px x
64-cout tile = 256 accumulator registers in AGPRs + 256 VGPRs (512 per lane), 32 x 16 px x 128 couts per workgroup: half the weight-staging
operations and 0.56 instead of 0.625 halo fragments per MFMA row against the default (flag 3 = 3: <2,2,0,8,4>, two workgroups per CU), but nothing
on a SIMD to cover a tile's ends.  Bare layer loops on random data, >= 2 s each (the chip settles at its power cap), outputs compared bit for bit.
   python tools/bench_one_wave_per_simd.py [seconds per case]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_tagger_amd import _lib
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
ctx = _lib.Context(0); dev = torch.device("cuda:0")
vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
for (B, H, W, Cin, Cout) in ((16, 256, 256, 512, 512), (16, 512, 512, 256, 256), (16, 1024, 1024, 128, 128)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, W, Cin, generator=g).to(dev, torch.bfloat16)
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) * (Cin * 9) ** -0.5).to(dev, torch.bfloat16)
    b = torch.zeros(Cout, device=dev)
    outs = {}
    for rep in range(2):
        for mode, name in ((3, "<2,2,0,8,4>  4 waves x 256 regs, two workgroups / CU (default)"), (0, "8 waves x 256 regs, one workgroup / CU"),
                           (4, "<2,2,0,16,6> 4 waves x 512 regs, ONE WAVE PER SIMD")):
            ctx.call("vt_set_flag", 3, mode)
            o = torch.empty(B, H, W, Cout, device=dev, dtype=torch.bfloat16)
            call = lambda: ctx.call("vt_op_conv2d", vp(x), vp(w), vp(b), None, None, vp(o), B, H, W, Cin, Cout, 3, 1, 1, 1, None)
            for _ in range(5): call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n, t0 = 0, time.perf_counter()
            e0.record()
            while time.perf_counter() - t0 < secs:
                for _ in range(20): call()
                n += 20
                torch.cuda.synchronize()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            outs[mode] = o
            same = "" if mode == 3 else f"  == default bit for bit: {torch.equal(o, outs[3])}"
            print(f"rep {rep} B{B} {H}x{W} {Cin}->{Cout}  {name:66s} {ms:7.3f} ms  {2.0 * B * H * W * Cout * 9 * Cin / ms / 1e9:7.1f} TFLOP/s{same}", flush=True)
ctx.call("vt_set_flag", 3, 3)
