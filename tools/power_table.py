"""Socket power and SMU clock per kind of work (hwmon power1_input / freq1_input of the card this process computes on, polled every 50 ms while a
loop of one kind of launch runs for ~2.5 s).  The cap is power1_cap (1400 W).  A loop that sits AT the cap is energy-limited: its rate is
(work per joule) x cap; a loop below the cap is limited by something else and shows what that kind of work costs.
   python tools/power_table.py            (needs tools/bin/mfma_peak: hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_peak tools/mfma_peak.hip)"""
import ctypes, glob, os, subprocess, sys, threading, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
from vae_tagger_amd import _lib

def read(p):
    try:
        return int(open(p).read().strip())
    except Exception:
        return None
hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
ctx = _lib.Context(0); dev = torch.device("cuda:0")
vp = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)

class Poll:
    def __init__(self, dirs): self.dirs = dirs; self.rows = []; self.stop = False
    def run(self):
        while not self.stop:
            self.rows.append((time.time(), [(read(d + "/power1_input"), read(d + "/freq1_input")) for d in self.dirs]))
            time.sleep(0.05)
    def __enter__(self):
        self.rows = []; self.stop = False; self.th = threading.Thread(target=self.run, daemon=True); self.th.start(); return self
    def __exit__(self, *a): self.stop = True; self.th.join()

# which card is ours: the one whose power moves when this process loads the GPU
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
with Poll(hw) as p0:
    time.sleep(0.5); t_load = time.time()
    for _ in range(300): a @ a
    torch.cuda.synchronize()
def med(v): v = sorted(x for x in v if x is not None); return v[len(v) // 2] if v else 0
before = [med([r[1][i][0] for r in p0.rows if r[0] < t_load]) for i in range(len(hw))]
after = [med([r[1][i][0] for r in p0.rows if r[0] > t_load + 0.3]) for i in range(len(hw))]
mine = max(range(len(hw)), key=lambda i: after[i] - before[i])
card = hw[mine]
print(f"card under this process: {card.split('/')[4]} (power {before[mine] / 1e6:.0f} -> {after[mine] / 1e6:.0f} W under a torch matmul loop); cap {read(card + '/power1_cap') / 1e6:.0f} W", flush=True)
del a

def report(label, rows, rate=""):
    rows = rows[len(rows) // 3:]                          # the clock and the power average settle within the first third
    pw = [r[1][0][0] for r in rows]; fq = [r[1][0][1] for r in rows]
    print(f"{label:64s} power median {med(pw) / 1e6:6.0f} W (max {max(x for x in pw if x) / 1e6:6.0f})   sclk median {med(fq) / 1e6:5.0f} MHz   {rate}", flush=True)

def loop(label, call, work_per_call, unit, seconds=2.5):
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); call(); e1.record(); torch.cuda.synchronize()
    n = max(3, int(seconds * 1e3 / max(e0.elapsed_time(e1), 1e-3)))
    with Poll([card]) as p:
        e0.record()
        for _ in range(n): call()
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    report(label, p.rows, f"{ms:8.3f} ms per call, {work_per_call / ms / 1e9:8.1f} {unit}")

with Poll([card]) as p:
    time.sleep(2.0)
report("idle (nothing queued)", p.rows)

# ---- bare MFMA loops (separate binary; samples attributed by the time each result line arrives)
exe = os.path.join(root, "tools", "bin", "mfma_peak")
if os.path.exists(exe):
    with Poll([card]) as p:
        child = subprocess.Popen([exe, "2.0"], stdout=subprocess.PIPE, text=True)
        marks = [(time.time(), None)]
        for line in child.stdout:
            marks.append((time.time(), line.strip()))
        child.wait()
    for (t0, _), (t1, line) in zip(marks, marks[1:]):
        rows = [r for r in p.rows if t0 + 0.2 < r[0] < t1]
        if rows and line and "1 wave" in line:
            report("bare " + " ".join(line.split()[:1] + line.split("SIMD")[1].split()[:1]), rows, line.split("sustained")[1].split("(")[0].strip())

# ---- library kernels
def conv_case(B, H, W, Cin, Cout, fill):
    torch.manual_seed(0)
    x = (torch.randn(B, H, W, Cin, device=dev) if fill == "random" else torch.zeros(B, H, W, Cin, device=dev)).to(torch.bfloat16)
    w = ((torch.randn(Cout, 3, 3, Cin, device=dev) * (Cin * 9) ** -0.5) if fill == "random" else torch.zeros(Cout, 3, 3, Cin, device=dev)).to(torch.bfloat16)
    b = torch.zeros(Cout, device=dev); o = torch.empty(B, H, W, Cout, device=dev, dtype=torch.bfloat16)
    loop(f"conv3x3_halo_kernel bf16 {Cin}->{Cout} @{H}^2 x{B}, {fill}", lambda: ctx.call("vt_op_conv2d", vp(x), vp(w), vp(b), None, None, vp(o), B, H, W, Cin, Cout, 3, 1, 1, 1, None),
         2.0 * B * H * W * Cout * 9 * Cin, "TFLOP/s")
for shape in ((16, 1024, 1024, 128, 128), (16, 512, 512, 256, 256), (16, 256, 256, 512, 512)):
    for fill in ("random", "zeros"):
        conv_case(*shape, fill)

def conv8_case(B, H, W, Cin, Cout):
    torch.manual_seed(0)
    x = torch.randn(B, H, W, Cin, device=dev); w = torch.randn(Cout, Cin, 3, 3, device=dev) * (Cin * 9) ** -0.5
    out = torch.empty(B, H, W, Cout, device=dev)
    n = ctx.lib.vt_op_conv3x3_fp8_workspace_bytes(B, H, W, Cin, Cout)
    ws = torch.empty(n + 256, device=dev, dtype=torch.uint8); ptr = (ws.data_ptr() + 255) // 256 * 256
    loop(f"vt_op_conv3x3_fp8 (quantise + conv3x3_halo_fp8_kernel) {Cin}->{Cout} @{H}^2 x{B}", lambda: ctx.call("vt_op_conv3x3_fp8", vp(x), vp(w), None, None, vp(out), B, H, W, Cin, Cout, 1, ctypes.c_void_p(ptr), None),
         2.0 * B * H * W * Cout * 9 * Cin, "TFLOP/s incl. the quantising passes")
conv8_case(16, 512, 512, 256, 256)
conv8_case(16, 256, 256, 512, 512)

def gn_case(B, HW, C):
    x = torch.randn(B, HW, C, device=dev).to(torch.bfloat16); y = torch.empty_like(x)
    g = torch.ones(C, device=dev); bt = torch.zeros(C, device=dev)
    ws = torch.empty(ctx.lib.vt_op_groupnorm_workspace_bytes(B, HW, C) + 256, device=dev, dtype=torch.uint8)
    loop(f"vt_op_groupnorm (statistics pass + apply pass) {C} ch @{HW} px x{B}", lambda: ctx.call("vt_op_groupnorm", vp(x), 1, B, HW, C, 32, ctypes.c_float(1e-6), vp(g), vp(bt), 1, vp(y), vp(ws), None),
         B * HW * C * 6.0, "GB/s (2 B read twice + 2 B written)")
try:
    gn_case(16, 1024 * 1024, 128)
except Exception as e:
    print("groupnorm case failed:", e)
src = torch.empty(1 << 30, device=dev, dtype=torch.uint8); dst = torch.empty_like(src)
loop("torch copy 1 GiB (HBM read + write)", lambda: dst.copy_(src), 2.0 * (1 << 30), "GB/s")
