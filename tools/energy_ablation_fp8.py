"""The fp8 halo kernel on the diagnostic builds of tools/energy_ablation.py (no fragment reads / no LDS-DMA staging / neither / weights from an
L1-sized window): kernel time from the library's own hipEvents (vt_op_conv3x3_fp8 quantises inside the call; only the conv launch is timed).
   python tools/energy_ablation_fp8.py LIB.so"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_tagger_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
ctx = _lib.Context(0); dev = torch.device("cuda:0")
vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
name = os.path.basename(sys.argv[1])
for (B, H, W, Cin, Cout) in ((16, 1024, 1024, 128, 128), (16, 512, 512, 256, 256), (16, 256, 256, 512, 512)):
    torch.manual_seed(0)
    x = torch.randn(B, H, W, Cin, device=dev); w = torch.randn(Cout, Cin, 3, 3, device=dev) * (Cin * 9) ** -0.5
    out = torch.empty(B, H, W, Cout, device=dev)
    n = ctx.lib.vt_op_conv3x3_fp8_workspace_bytes(B, H, W, Cin, Cout)
    ws = torch.empty(n + 256, device=dev, dtype=torch.uint8); ptr = (ws.data_ptr() + 255) // 256 * 256
    call = lambda: ctx.call("vt_op_conv3x3_fp8", vp(x), vp(w), None, None, vp(out), B, H, W, Cin, Cout, 1, ctypes.c_void_p(ptr), None)
    for _ in range(20): call()
    torch.cuda.synchronize()
    ns = ctx.lib.vt_profile_num_configs()
    la = (ctypes.c_longlong * ns)(); ms = (ctypes.c_double * ns)(); fl = (ctypes.c_double * ns)(); nm = (ctypes.c_char_p * ns)()
    ctx.call("vt_profile_begin")
    for _ in range(60): call()
    ctx.call("vt_profile_end", ns, la, ms, fl, nm)
    i = max((k for k in range(ns) if nm[k] and b"halo_fp8" in nm[k]), key=lambda k: la[k])       # (Cin <= 128 launches have a slot of their own)
    print(f"{name:22s} fp8 {Cin:4d}->{Cout:4d} @{H:4d}^2: {ms[i] / la[i]:7.3f} ms {fl[i] / ms[i] / 1e9:7.1f} TFLOP/s", flush=True)
    del x, w, out, ws
