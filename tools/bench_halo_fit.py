"""Separate the halo conv's per-K-step cost from its per-tile overhead: time ~ tiles_per_CU * (overhead + nk * step).
   python tools/bench_halo_fit.py [lib.so]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_tagger_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
ctx = _lib.Context(0); dev = torch.device("cuda:0")
vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
print("lib:", _lib.LIB_PATH, flush=True)
def run(B, H, W, Cin, Cout, occ2, out="bf16", iters=10):
    ctx.call("vt_set_flag", 3, occ2)
    torch.manual_seed(0)
    x = torch.randn(B, H, W, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) * (Cin * 9) ** -0.5).to(torch.bfloat16)
    b = torch.zeros(Cout, device=dev)
    o16 = torch.empty(B, H, W, Cout, device=dev, dtype=torch.bfloat16)
    o32 = torch.empty(B, H, W, Cout, device=dev) if out == "f32" else None
    def call():
        ctx.call("vt_op_conv2d", vp(x), vp(w), vp(b), None, vp(o32), None if o32 is not None else vp(o16), B, H, W, Cin, Cout, 3, 1, 1, 1, None)
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * B * H * W * Cout * 9 * Cin
    rows = 16 if (Cout % 256 == 0 or occ2) else 32
    bc = 128 if (occ2 == 3 or Cout % 256) else 256
    tiles = B * (H // rows) * (W // 16) * (Cout // bc)
    per_cu = tiles / 256.0 / (2 if (occ2 == 3 or (Cout % 256 and occ2)) else 1)
    us_tile = ms * 1e3 / per_cu
    print(f"B{B} {H}x{W} {Cin:4d}->{Cout} occ2={occ2} {out}: {ms:7.3f} ms {fl/ms/1e9:7.1f} TF/s  nk={Cin//32*9:4d} tiles/CU-slot {per_cu:6.1f}  us/tile {us_tile:7.2f}", flush=True)
    return us_tile
import numpy as np
for Cout, shape in ((128, (8, 1024, 1024)), (256, (16, 512, 512))):
    for occ2 in (3, 0):
        pts = []
        for Cin in (32, 64, 128, 256, 512, 1024):
            print(f"occ2 {occ2}: ", end="")
            pts.append((Cin // 32 * 9, run(shape[0], shape[1], shape[2], Cin, Cout, occ2, iters=5)))
        # least-squares fit us/tile-slot = overhead + nk * step over the four largest K
        A = np.array([[1.0, p[0]] for p in pts[2:]]); y = np.array([p[1] for p in pts[2:]])
        (ov, st), *_ = np.linalg.lstsq(A, y, rcond=None)
        bc = 128 if (occ2 == 3 or Cout % 256) else 256
        slots = 2 if occ2 == 3 or (Cout % 256 and occ2) else 1
        print(f"   fit Cout {Cout} occ2 {occ2}: overhead {ov:6.2f} us/tile-slot, step {st:6.3f} us  (tile {bc} couts, {slots} slot(s)/CU: "
              f"{256 * bc * 32 * 2 / st / 1e6 * slots * 256 / 1e6:6.1f} TF/s asymptotic)", flush=True)
ctx.call("vt_set_flag", 3, 3)
