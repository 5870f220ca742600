"""CU-masked streams: does the power-bound halo conv lose less than proportionally on fewer CUs, and what does the HBM-bound
GroupNorm pass reach on the rest?  (the one overlap experiment not yet run.)
   python tools/cu_mask.py [stamp-lib.so]
Streams come from hipExtStreamCreateWithCUMask; with the diagnostic build as argument the mask's bit -> (XCC, CU) mapping is read
back from HW_ID stamps.  Reported per split N | 256 - N: the conv alone on N CUs, the GroupNorm pass alone on 256 - N CUs, and
both at once (each timed with hipEvents on its own stream while the other stream is kept busy for longer)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_tagger_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
ctx = _lib.Context(0); dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def masked_stream(bits):
    """bits: iterable of CU bit indices (0..255) that are enabled"""
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return torch.cuda.ExternalStream(s.value, device=dev), s


torch.manual_seed(0)
# conv operands: the two halo-kernel shapes that make most of the step
SHAPES = {"256->256 @512^2": (16, 512, 512, 256, 256), "128->128 @1024^2": (16, 1024, 1024, 128, 128)}
conv = {}
for name, (B, H, W, Cin, Cout) in SHAPES.items():
    x = torch.randn(B, H, W, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) * (Cin * 9) ** -0.5).to(torch.bfloat16)
    b = torch.zeros(Cout, device=dev)
    o = torch.empty(B, H, W, Cout, device=dev, dtype=torch.bfloat16)
    conv[name] = (x, w, b, o, 2.0 * B * H * W * Cout * 9 * Cin)
# GroupNorm pass operands: the 128-channel 1024^2 tensor (fp16 in, bf16 out)
gB, gHW, gC = 16, 1024 * 1024, 128
gx = torch.randn(gB, gHW, gC, device=dev).to(torch.float16)
gy = torch.empty(gB, gHW, gC, device=dev, dtype=torch.bfloat16)
gam = torch.ones(gC, device=dev); bet = torch.zeros(gC, device=dev)
gws = torch.empty(ctx.lib.vt_op_groupnorm_workspace_bytes(gB, gHW, gC) + 256, dtype=torch.uint8, device=dev)
gbytes = gB * gHW * gC * (2 + 2 + 2)                     # stats read + apply read + apply write


def run_conv(name, stream):
    x, w, b, o, _ = conv[name]
    B, H, W, Cin, Cout = SHAPES[name]
    ctx.call("vt_op_conv2d", vp(x), vp(w), vp(b), None, None, vp(o), B, H, W, Cin, Cout, 3, 1, 1, 1, ctypes.c_void_p(stream.cuda_stream))


def run_gn(stream):
    ctx.call("vt_op_groupnorm", vp(gx), 2, gB, gHW, gC, 32, 1e-6, vp(gam), vp(bet), 1, vp(gy), vp(gws), ctypes.c_void_p(stream.cuda_stream))


def timed(fn, stream, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(n):
        fn()
    e1.record(stream)
    return e0, e1


def census(stream, tag):
    if not hasattr(ctx.lib, "vt_debug_halo_stamps"):
        return
    ctx.lib.vt_debug_halo_stamps.argtypes = [ctypes.c_void_p]; ctx.lib.vt_debug_halo_stamps.restype = ctypes.c_int
    st = torch.zeros(1 << 16, 16, dtype=torch.int64, device=dev)
    assert ctx.lib.vt_debug_halo_stamps(st.data_ptr()) == 0
    run_conv("256->256 @512^2", stream); torch.cuda.synchronize()
    assert ctx.lib.vt_debug_halo_stamps(None) == 0
    s = st.cpu(); s = s[s[:, 6] > 0]
    xcc = (s[:, 7] >> 32); cu = (s[:, 7] & 0xffffffff) >> 8 & 0xff
    per = [len(torch.unique(cu[xcc == k])) for k in range(8)]
    print(f"   census {tag}: CUs used per XCC {per} (total {sum(per)})", flush=True)


full, _ = masked_stream(range(256))
for name in SHAPES:
    for _ in range(3):
        run_conv(name, full)
for _ in range(2):
    run_gn(full)
torch.cuda.synchronize()
census(full, "all 256 bits")
base = {}
for name in SHAPES:
    e0, e1 = timed(lambda: run_conv(name, full), full, 10); torch.cuda.synchronize()
    base[name] = e0.elapsed_time(e1) / 10
    print(f"conv {name} on 256 CUs: {base[name]:.3f} ms  {conv[name][4] / base[name] / 1e9:.0f} TF/s")
e0, e1 = timed(lambda: run_gn(full), full, 5); torch.cuda.synchronize()
gbase = e0.elapsed_time(e1) / 5
print(f"GroupNorm pass (stats + apply, {gbytes / 1e9:.1f} GB) on 256 CUs: {gbase:.3f} ms  {gbytes / gbase / 1e6:.0f} GB/s", flush=True)

for N in (224, 208, 192, 160, 128):
    sa, _ = masked_stream(range(N))                    # low N bits: N / 8 CUs of every XCC if bit i belongs to XCC i % 8
    sb, _ = masked_stream(range(N, 256))
    print(f"split {N} | {256 - N}:", flush=True)
    census(sa, f"bits 0..{N - 1}")
    census(sb, f"bits {N}..255")
    for name in SHAPES:
        for _ in range(2):
            run_conv(name, sa)
        run_gn(sb)
        torch.cuda.synchronize()
        e0, e1 = timed(lambda: run_conv(name, sa), sa, 10); torch.cuda.synchronize()
        alone = e0.elapsed_time(e1) / 10
        g0, g1 = timed(lambda: run_gn(sb), sb, 3); torch.cuda.synchronize()
        galone = g0.elapsed_time(g1) / 3
        # both at once: the GroupNorm stream is given more work than the conv stream needs time, and vice versa
        ng = max(2, int(alone * 10 / galone) + 2)
        g0, g1 = timed(lambda: run_gn(sb), sb, ng)
        e0, e1 = timed(lambda: run_conv(name, sa), sa, 10)
        torch.cuda.synchronize()
        beside = e0.elapsed_time(e1) / 10
        nc = max(2, int(galone * 3 / alone) + 4)
        e0, e1 = timed(lambda: run_conv(name, sa), sa, nc)
        g0, g1 = timed(lambda: run_gn(sb), sb, 3)
        torch.cuda.synchronize()
        gbeside = g0.elapsed_time(g1) / 3
        print(f"   conv {name}: alone on {N} CUs {alone:.3f} ms (x{alone / base[name]:.3f} of 256 CUs; proportional would be x{256 / N:.3f}); "
              f"beside the GroupNorm stream {beside:.3f} ms (x{beside / base[name]:.3f})")
        print(f"   GroupNorm pass on {256 - N} CUs: alone {galone:.3f} ms ({gbytes / galone / 1e6:.0f} GB/s), beside the conv {gbeside:.3f} ms ({gbytes / gbeside / 1e6:.0f} GB/s); "
              f"serial conv + GN on 256 CUs = {base[name] + gbase:.3f} ms, overlapped = max({beside:.3f}, {gbeside:.3f})", flush=True)
