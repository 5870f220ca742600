"""Where does the bf16 halo kernel's energy go?  The same launches (random operands, batch 16) on diagnostic builds that leave parts of the K-loop out
(results are wrong by construction; only time, power and clock are read):
   base                      the product kernel
   -DEXP_NO_FRAG_READS       no LDS -> register fragment reads after step 0 (the matrix pipe keeps multiplying the step-0 operands)
   -DEXP_NO_DMA              no L2 -> LDS staging after the first two halos / four weight tiles
   both                      MFMAs + prologue / epilogue (HBM output, residual-free) only
   -DW_WINDOW_EXPERIMENT=2   all weight tiles from a 16-KB window (L1-resident): the L2 -> L1 part of the weight staging
   python tools/energy_ablation.py LIB.so      (one library per process; tools/build_variant.sh builds them)"""
import ctypes, glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_tagger_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
ctx = _lib.Context(0); dev = torch.device("cuda:0")
vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
def read(p):
    try: return int(open(p).read().strip())
    except Exception: return None
hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
class Poll:
    def __init__(self, dirs): self.dirs = dirs
    def run(self):
        while not self.stop:
            self.rows.append((time.time(), [(read(d + "/power1_input"), read(d + "/freq1_input")) for d in self.dirs])); time.sleep(0.05)
    def __enter__(self): self.rows = []; self.stop = False; self.th = threading.Thread(target=self.run, daemon=True); self.th.start(); return self
    def __exit__(self, *a): self.stop = True; self.th.join()
def med(v): v = sorted(x for x in v if x is not None); return v[len(v) // 2] if v else 0
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
with Poll(hw) as p0:
    time.sleep(0.4); t_load = time.time()
    for _ in range(200): a @ a
    torch.cuda.synchronize()
mine = max(range(len(hw)), key=lambda i: med([r[1][i][0] for r in p0.rows if r[0] > t_load + 0.2]) - med([r[1][i][0] for r in p0.rows if r[0] < t_load]))
card = hw[mine]; del a
name = os.path.basename(sys.argv[1])
for (B, H, W, Cin, Cout) in ((16, 1024, 1024, 128, 128), (16, 512, 512, 256, 256), (16, 256, 256, 512, 512)):
    torch.manual_seed(0)
    x = torch.randn(B, H, W, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) * (Cin * 9) ** -0.5).to(torch.bfloat16)
    b = torch.zeros(Cout, device=dev); o = torch.empty(B, H, W, Cout, device=dev, dtype=torch.bfloat16)
    call = lambda: ctx.call("vt_op_conv2d", vp(x), vp(w), vp(b), None, None, vp(o), B, H, W, Cin, Cout, 3, 1, 1, 1, None)
    for _ in range(5): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 600
    with Poll([card]) as p:
        e0.record()
        for _ in range(n): call()
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    rows = p.rows[len(p.rows) // 3:]
    print(f"{name:22s} {Cin:4d}->{Cout:4d} @{H:4d}^2: {ms:7.3f} ms {2.0 * B * H * W * Cout * 9 * Cin / ms / 1e9:7.1f} TFLOP/s   power {med([r[1][0][0] for r in rows]) / 1e6:6.0f} W   sclk {med([r[1][0][1] for r in rows]) / 1e6:5.0f} MHz", flush=True)
