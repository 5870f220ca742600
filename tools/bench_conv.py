"""Time single conv layers through the op-level C ABI (experiments / ablations).  Usage:
   python tools/bench_conv.py [path/to/lib.so]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_tagger_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
ctx = _lib.Context(0)
dev = torch.device("cuda:0")
vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
def run(B, H, W, Cin, Cout, mode, iters=10):
    g = torch.Generator(device="cpu").manual_seed(0)
    x32 = torch.randn(B, H, W, Cin, generator=g).to(dev)
    x16 = x32.to(torch.bfloat16)
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) * (Cin * 9) ** -0.5).to(dev, torch.bfloat16)
    b = torch.zeros(Cout, device=dev)
    ss = torch.stack([torch.ones(B, Cin), torch.zeros(B, Cin)], -1).to(dev).contiguous()
    res = torch.randn(B, H, W, Cout, device=dev)
    o32 = torch.empty(B, H, W, Cout, device=dev)
    o16 = torch.empty(B, H, W, Cout, device=dev, dtype=torch.bfloat16)
    def call():
        if mode == "raw":      # XT=0, bf16 out
            ctx.call("vt_op_conv2d", vp(x16), vp(w), vp(b), None, None, vp(o16), B, H, W, Cin, Cout, 3, 1, 1, 1, None)
        elif mode == "raw_res":  # XT=0, fp32 out + residual
            ctx.call("vt_op_conv2d", vp(x16), vp(w), vp(b), vp(res), vp(o32), None, B, H, W, Cin, Cout, 3, 1, 1, 1, None)
        elif mode == "f32norm":  # XT=1 -> bf16 out (conv1)
            ctx.call("vt_op_norm_silu_conv3x3", vp(x32), 0, vp(ss), vp(w), vp(b), None, None, vp(o16), B, H, W, Cin, Cout, None)
        elif mode == "bf16norm_res":  # XT=2 -> fp32 out + residual (conv2)
            ctx.call("vt_op_norm_silu_conv3x3", vp(x16), 1, vp(ss), vp(w), vp(b), vp(res), vp(o32), None, B, H, W, Cin, Cout, None)
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * B * H * W * Cout * 9 * Cin
    print(f"  {mode:14s} B{B} {H}x{W} {Cin}->{Cout}: {ms:7.3f} ms  {fl/ms/1e9:7.1f} TFLOP/s", flush=True)
print("lib:", _lib.LIB_PATH)
for shape in [(8, 1024, 1024, 128, 128), (8, 512, 512, 256, 256), (8, 256, 256, 512, 512)]:
    for mode in ("raw", "f32norm", "bf16norm_res"):
        run(*shape, mode)
