"""Which LANE's (sum, sum of squares, pivot, valid mask) differs run to run in the halo conv's GroupNorm epilogue?
Needs a -DGNIL_DUMP build (tools/build_variant.sh dump "-DGNIL_DUMP"):
   VAE_TAGGER_HIP_LIB=vae_tagger_amd/csrc/exp/libvt_dump.so python tests/diagnostics/halo_partials_lane_dump.py [reps]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vae_tagger_amd import _lib
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx = _lib.Context(0); dev = torch.device("cuda:0")
vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
B, H, W, Cin, Cout = 4, 264, 136, 128, 128
g = torch.Generator().manual_seed(0)
x = torch.randn(B, H, W, Cin, generator=g).to(dev, torch.bfloat16)
w = (torch.randn(Cout, 3, 3, Cin, generator=g) * (Cin * 9) ** -0.5).to(dev, torch.bfloat16)
b = torch.randn(Cout, generator=g).to(dev); gam = torch.ones(Cout, device=dev); bet = torch.zeros(Cout, device=dev)
o16 = torch.empty(B, H, W, Cout, device=dev, dtype=torch.bfloat16)
n = ctx.lib.vt_op_conv2d_gn_workspace_bytes(B, H, W, Cout)
ws = torch.zeros(n // 4 + 64, device=dev)
ss = torch.zeros(B, Cout, 2, device=dev)
tx = (W + 15) // 16; tiles = tx * ((H + 15) // 16)
npart = B * tiles * 32 * 3
nblk = B * tiles
dump = torch.zeros(nblk * 256 * 4 * 4, device=dev)
fn = ctx.lib.vt_debug_gnil_dump; fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
assert fn(vp(dump)) == 0
def run():
    ctx.call("vt_op_conv2d_gn", vp(x), vp(w), vp(b), None, None, vp(o16), B, H, W, Cin, Cout, 3, 1, 1, 1, 32, 1e-6, vp(gam), vp(bet), vp(ss), vp(ws), None)
run(); torch.cuda.synchronize()
ref = ws[:npart].clone(); rd = dump.clone()
bad = 0; shown = 0
for rep in range(reps):
    run()
    cur = ws[:npart]
    pdiff = not torch.equal(cur.view(torch.int32), ref.view(torch.int32))
    ddiff = not torch.equal(dump.view(torch.int32), rd.view(torch.int32))
    if pdiff or ddiff:
        bad += 1
        if shown < 6:
            shown += 1
            d = torch.nonzero(dump.view(torch.int32) != rd.view(torch.int32)).flatten()
            rows = []
            for i in d[:4].tolist():
                comp = i % 4; q = (i // 4) % 4; tid = (i // 16) % 256; blk = i // (16 * 256)
                a0, a1 = rd[i], dump[i]
                a0, a1 = f"{a0.item():.6g}", f"{a1.item():.6g}"
                rows.append(f"(blk {blk}, wave {tid >> 6}, lane {tid & 63} (fq {(tid >> 4) & 3}, fr {tid & 15}), q {q}, {'s ss piv v00'.split()[comp]}: {a0} -> {a1})")
            # the pivot each differing row must have used: s = sum(v) - m * pivot, m values per lane -> pivot shift = -(s' - s) / m
            i0 = d[0].item(); q0 = (i0 // 4) % 4; t0 = (i0 // 16) % 256; b0 = i0 // 4096; row = (t0 >> 4) << 4
            base = lambda t, qq, c: ((b0 * 256 + t) * 4 + qq) * 4 + c
            print(f"      blk {b0} wave {t0 >> 6} fq {(t0 >> 4) & 3} q {q0}: pivot dumped {rd[base(row, q0, 2)].item():.6g} -> {dump[base(row, q0, 2)].item():.6g};"
                  f" delta s per lane {[round((dump[base(row + k, q0, 0)] - rd[base(row + k, q0, 0)]).item(), 4) for k in range(16)]};"
                  f" v[q][0][0] of the row's lanes {[round(dump[base(row + k, q0, 3)].item(), 4) for k in range(16)]};"
                  f" v[.][0][0] of lane 48 for q 0..3 {[round(dump[base(row, qq, 3)].item(), 4) for qq in range(4)]};"
                  f" pivots of the row for q 0..3 {[round(dump[base(row, qq, 2)].item(), 4) for qq in range(4)]}", flush=True)
            dp = torch.nonzero(cur.view(torch.int32) != ref.view(torch.int32)).flatten()[:4].tolist()
            print(f"   rep {rep}: partial words differing {dp} (tile, group) = {[((i // 96) % tiles, (i // 3) % 32) for i in dp]}; {d.numel()} dump words differ: " + " ".join(rows), flush=True)
print(f"B{B} {H}x{W}: differ in {bad}/{reps} runs", flush=True)
