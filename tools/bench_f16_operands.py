"""Do fp16 MFMA operands cost speed against bf16 on this chip (same cycles per instruction; the question is the clock the chip
holds)?  The same halo-conv launch on bf16 data (shipping build) and on fp16 data with v_mfma_f32_16x16x32_f16 (build with
-DHALO_F16: tools/build_variant.sh f16 "-DHALO_F16").   python tools/bench_f16_operands.py"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_tagger_amd import _lib
f16 = len(sys.argv) > 1 and sys.argv[1] == "f16"
if f16:
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "exp", "libvt_f16.so")
ctx = _lib.Context(0); dev = torch.device("cuda:0")
vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
dt = torch.float16 if f16 else torch.bfloat16
for (B, H, W, Cin, Cout) in ((16, 512, 512, 256, 256), (16, 1024, 1024, 128, 128), (16, 256, 256, 512, 512)):
    torch.manual_seed(0)
    x = torch.randn(B, H, W, Cin, device=dev).to(dt)
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) * (Cin * 9) ** -0.5).to(dt)
    b = torch.zeros(Cout, device=dev)
    o = torch.empty(B, H, W, Cout, device=dev, dtype=torch.bfloat16)
    call = lambda: ctx.call("vt_op_conv2d", vp(x), vp(w), vp(b), None, None, vp(o), B, H, W, Cin, Cout, 3, 1, 1, 1, None)
    for _ in range(10): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{'fp16' if f16 else 'bf16'} operands B{B} {H}x{W} {Cin}->{Cout}: {ms:7.3f} ms  {2.0 * B * H * W * Cout * 9 * Cin / ms / 1e9:7.1f} TFLOP/s", flush=True)
