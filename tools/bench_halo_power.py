"""Is the halo conv limited by issue / latency or by the chip's power management?  The same launch on random operands, on
all-zero operands (same instruction stream and memory traffic, no data toggling in the matrix pipe) and on constant operands.
   python tools/bench_halo_power.py"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_tagger_amd import _lib
ctx = _lib.Context(0); dev = torch.device("cuda:0")
vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
def run(B, H, W, Cin, Cout, fill, iters=20):
    torch.manual_seed(0)
    if fill == "random":
        x = torch.randn(B, H, W, Cin, device=dev).to(torch.bfloat16)
        w = (torch.randn(Cout, 3, 3, Cin, device=dev) * (Cin * 9) ** -0.5).to(torch.bfloat16)
    elif fill == "zeros":
        x = torch.zeros(B, H, W, Cin, device=dev, dtype=torch.bfloat16)
        w = torch.zeros(Cout, 3, 3, Cin, device=dev, dtype=torch.bfloat16)
    else:
        x = torch.full((B, H, W, Cin), 1.0, device=dev, dtype=torch.bfloat16)
        w = torch.full((Cout, 3, 3, Cin), 0.5, device=dev, dtype=torch.bfloat16)
    b = torch.zeros(Cout, device=dev)
    o16 = torch.empty(B, H, W, Cout, device=dev, dtype=torch.bfloat16)
    def call():
        ctx.call("vt_op_conv2d", vp(x), vp(w), vp(b), None, None, vp(o16), B, H, W, Cin, Cout, 3, 1, 1, 1, None)
    for _ in range(10): call()                       # let the clocks settle under this load
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * B * H * W * Cout * 9 * Cin
    print(f"{fill:8s} B{B} {H}x{W} {Cin}->{Cout}: {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)
for shape in ((16, 512, 512, 256, 256), (16, 1024, 1024, 128, 128), (16, 512, 512, 1024, 256)):
    for fill in ("random", "zeros", "const", "random"):
        run(*shape, fill)

# the fp8 kernel (vt_op_conv3x3_fp8 quantises inside the call; the kernel itself is timed by the library's hipEvents, slot "conv3x3_halo_fp8_kernel")
def run8(B, H, W, Cin, Cout, fill, iters=10):
    torch.manual_seed(0)
    if fill == "random":
        x = torch.randn(B, H, W, Cin, device=dev); w = torch.randn(Cout, Cin, 3, 3, device=dev) * (Cin * 9) ** -0.5
    elif fill == "zeros":
        x = torch.zeros(B, H, W, Cin, device=dev); w = torch.zeros(Cout, Cin, 3, 3, device=dev)
    else:
        x = torch.full((B, H, W, Cin), 1.0, device=dev); w = torch.full((Cout, Cin, 3, 3), 0.5, device=dev)
    out = torch.empty(B, H, W, Cout, device=dev)
    n = ctx.lib.vt_op_conv3x3_fp8_workspace_bytes(B, H, W, Cin, Cout)
    ws = torch.empty(n + 256, device=dev, dtype=torch.uint8)
    ptr = (ws.data_ptr() + 255) // 256 * 256
    def call():
        ctx.call("vt_op_conv3x3_fp8", vp(x), vp(w), None, None, vp(out), B, H, W, Cin, Cout, 1, ctypes.c_void_p(ptr), None)
    for _ in range(10): call()
    torch.cuda.synchronize()
    ns = ctx.lib.vt_profile_num_configs()
    la = (ctypes.c_longlong * ns)(); ms = (ctypes.c_double * ns)(); fl = (ctypes.c_double * ns)(); nm = (ctypes.c_char_p * ns)()
    ctx.call("vt_profile_begin")
    for _ in range(iters): call()
    ctx.call("vt_profile_end", ns, la, ms, fl, nm)
    i = max((k for k in range(ns) if nm[k] and b"fp8" in nm[k]), key=lambda k: la[k])       # (Cin <= 128 launches have a slot of their own)
    print(f"fp8 {fill:8s} B{B} {H}x{W} {Cin}->{Cout}: {ms[i] / la[i]:7.3f} ms  {fl[i] / ms[i] / 1e9:7.1f} TFLOP/s", flush=True)
for shape in ((16, 512, 512, 256, 256), (8, 1024, 1024, 128, 128), (16, 256, 256, 512, 512), (8, 512, 512, 1024, 256)):
    for fill in ("random", "zeros", "random"):
        run8(*shape, fill)
