"""Raw GroupNorm partials of the halo conv's epilogue, run to run: which (image, tile, group, component) differ?
   python tests/diagnostics/halo_partials_diff.py [reps]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vae_tagger_amd import _lib
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx = _lib.Context(0); dev = torch.device("cuda:0")
vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
for (B, H, W, Cin, Cout) in ((4, 264, 136, 128, 128), (4, 264, 136, 128, 128), (2, 520, 264, 128, 128), (4, 256, 128, 128, 128)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, W, Cin, generator=g).to(dev, torch.bfloat16)
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) * (Cin * 9) ** -0.5).to(dev, torch.bfloat16)
    b = torch.randn(Cout, generator=g).to(dev); gam = torch.ones(Cout, device=dev); bet = torch.zeros(Cout, device=dev)
    o16 = torch.empty(B, H, W, Cout, device=dev, dtype=torch.bfloat16)
    n = ctx.lib.vt_op_conv2d_gn_workspace_bytes(B, H, W, Cout)
    ws = torch.zeros(n // 4 + 64, device=dev)
    ss = torch.zeros(B, Cout, 2, device=dev)
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    npart = B * tiles * 32 * 3
    def run():
        ctx.call("vt_op_conv2d_gn", vp(x), vp(w), vp(b), None, None, vp(o16), B, H, W, Cin, Cout, 3, 1, 1, 1, 32, 1e-6, vp(gam), vp(bet), vp(ss), vp(ws), None)
    run(); torch.cuda.synchronize()
    ref = ws[:npart].clone(); ro = o16.clone()
    bad = 0; shown = 0
    for rep in range(reps):
        run()
        cur = ws[:npart]
        if not torch.equal(cur.view(torch.int32), ref.view(torch.int32)):
            bad += 1
            if shown < 3:
                shown += 1
                d = torch.nonzero(cur.view(torch.int32) != ref.view(torch.int32)).flatten()
                rows = []
                for i in d[:6].tolist():
                    comp = i % 3; grp = (i // 3) % 32; tile = (i // 96) % tiles; img = i // (96 * tiles)
                    rows.append(f"(img {img}, tile {tile} = ({tile // ((W + 15) // 16)}, {tile % ((W + 15) // 16)}), group {grp}, {'n mean M2'.split()[comp]}: {ref[i].item():.6g} -> {cur[i].item():.6g})")
                print(f"   rep {rep}: {d.numel()} words differ, outputs equal: {torch.equal(o16, ro)}: " + " ".join(rows), flush=True)
    print(f"B{B} {H}x{W} {Cin}->{Cout} ({tiles} tiles of 16x16): partials differ in {bad}/{reps} runs", flush=True)
