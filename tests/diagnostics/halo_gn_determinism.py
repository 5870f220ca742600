"""conv3x3 + GroupNorm-partials epilogue + finalize through vt_op_conv2d_gn at the layer shapes of a 136 x 264 image (ragged tiles at every level):
is (output, scale/shift) identical run to run, per halo tile mode?   python tests/diagnostics/halo_gn_determinism.py [reps]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vae_tagger_amd import _lib
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ctx = _lib.Context(0); dev = torch.device("cuda:0")
vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
for (B, H, W, Cin, Cout) in ((2, 33, 17, 512, 512), (2, 66, 34, 512, 512), (2, 66, 34, 256, 512), (2, 132, 68, 256, 256), (2, 132, 68, 128, 256), (2, 264, 136, 128, 128), (2, 32, 16, 512, 512)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, W, Cin, generator=g).to(dev, torch.bfloat16)
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) * (Cin * 9) ** -0.5).to(dev, torch.bfloat16)
    b = torch.randn(Cout, generator=g).to(dev); gam = torch.ones(Cout, device=dev); bet = torch.zeros(Cout, device=dev)
    res = torch.randn(B, H, W, Cout, generator=g).to(dev)
    o32 = torch.empty(B, H, W, Cout, device=dev)
    n = ctx.lib.vt_op_conv2d_gn_workspace_bytes(B, H, W, Cout)
    ws = torch.zeros(n // 4 + 64, device=dev)
    ss = torch.zeros(B, Cout, 2, device=dev)
    for occ2 in (3, 0):
        ctx.call("vt_set_flag", 3, occ2)
        def run():
            ctx.call("vt_op_conv2d_gn", vp(x), vp(w), vp(b), vp(res), vp(o32), None, B, H, W, Cin, Cout, 3, 1, 1, 1, 32, 1e-6, vp(gam), vp(bet), vp(ss), vp(ws), None)
        run(); torch.cuda.synchronize()
        ro, rs = o32.clone(), ss.clone()
        bad_o = bad_s = 0; worst = 0.0
        for rep in range(reps):
            run()
            if not torch.equal(o32, ro): bad_o += 1
            if not torch.equal(ss, rs):
                bad_s += 1; worst = max(worst, ((ss - rs).abs() / (rs.abs() + 1e-6)).max().item())
        print(f"B{B} {H}x{W} {Cin}->{Cout} flag3={occ2}: outputs differ in {bad_o}/{reps} runs, scale/shift in {bad_s}/{reps} (max rel {worst:.2e})", flush=True)
ctx.call("vt_set_flag", 3, 3)
