"""fp8 halo conv: per-K-step cost vs per-tile overhead, time per tile-slot ~ overhead + nk * step (nk = 9 * Cin / 64).
   python tools/bench_halo_fp8_fit.py"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vae_tagger_amd import _lib
ctx = _lib.Context(0); dev = torch.device("cuda:0")
vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
def run8(B, H, W, Cin, Cout, iters=8):
    torch.manual_seed(0)
    x = torch.randn(B, H, W, Cin, device=dev); w = torch.randn(Cout, Cin, 3, 3, device=dev) * (Cin * 9) ** -0.5
    out = torch.empty(B, H, W, Cout, device=dev)
    n = ctx.lib.vt_op_conv3x3_fp8_workspace_bytes(B, H, W, Cin, Cout)
    ws = torch.empty(n + 256, device=dev, dtype=torch.uint8)
    ptr = (ws.data_ptr() + 255) // 256 * 256
    def call():
        ctx.call("vt_op_conv3x3_fp8", vp(x), vp(w), None, None, vp(out), B, H, W, Cin, Cout, 1, ctypes.c_void_p(ptr), None)
    for _ in range(4): call()
    torch.cuda.synchronize()
    ns = ctx.lib.vt_profile_num_configs()
    la = (ctypes.c_longlong * ns)(); ms = (ctypes.c_double * ns)(); fl = (ctypes.c_double * ns)(); nm = (ctypes.c_char_p * ns)()
    ctx.call("vt_profile_begin")
    for _ in range(iters): call()
    ctx.call("vt_profile_end", ns, la, ms, fl, nm)
    i = max((k for k in range(ns) if nm[k] and b"halo_fp8" in nm[k]), key=lambda k: la[k])       # (Cin <= 128 launches have a slot of their own)
    t = ms[i] / la[i]
    tiles = B * (H // 8) * (W // 32) * (Cout // 128)
    us_tile = t * 1e3 / (tiles / 512.0)                       # two workgroups per CU
    print(f"fp8 B{B} {H}x{W} {Cin:4d}->{Cout}: {t:7.3f} ms {fl[i] / ms[i] / 1e9:7.1f} TF/s  nk={Cin // 64 * 9:4d}  us/tile-slot {us_tile:7.2f}", flush=True)
    return us_tile
for Cout, shape in ((128, (8, 1024, 1024)), (256, (16, 512, 512))):
    pts = [(Cin // 64 * 9, run8(*shape, Cin, Cout)) for Cin in (64, 128, 256, 512, 1024)]
    A = np.array([[1.0, p[0]] for p in pts[1:]]); y = np.array([p[1] for p in pts[1:]])
    (ov, st), *_ = np.linalg.lstsq(A, y, rcond=None)
    print(f"   fit Cout {Cout}: overhead {ov:6.2f} us/tile-slot, step {st:6.3f} us -> {256 * 128 * 64 * 2 / st / 1e6 * 512 / 1e6:6.1f} TF/s asymptotic", flush=True)
