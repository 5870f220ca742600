"""Does a consumer pass run faster / cheaper when the tensor its producer just wrote still sits in the 256-MB Infinity Cache?  producer: y <- x (reads x, writes y);
consumer: z <- y (reads y, writes z), timed alone with events, for tensor sizes around the cache size.  Also the library's GroupNorm apply (bf16 -> bf16) on the
output of a preceding pass of the same size.   python tools/mall_probe.py"""
import ctypes, glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
def read(p):
    try: return int(open(p).read().strip())
    except Exception: return None
hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
class Poll:
    def __init__(self, dirs): self.dirs = dirs
    def run(self):
        while not self.stop:
            self.rows.append([read(d + "/power1_input") for d in self.dirs]); time.sleep(0.05)
    def __enter__(self): self.rows = []; self.stop = False; self.th = threading.Thread(target=self.run, daemon=True); self.th.start(); return self
    def __exit__(self, *a): self.stop = True; self.th.join()
def med(v): v = sorted(x for x in v if x is not None); return v[len(v) // 2] if v else 0
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
idle = [read(d + "/power1_input") or 0 for d in hw]
with Poll(hw) as p0:
    for _ in range(200): a @ a
    torch.cuda.synchronize()
mine = max(range(len(hw)), key=lambda i: med([r[i] for r in p0.rows]) - idle[i]); card = hw[mine]; del a
for mb in (32, 64, 96, 128, 192, 256, 512, 2048):
    n = mb << 20
    x = torch.randint(0, 255, (n,), device=dev, dtype=torch.uint8); y = torch.empty_like(x); z = torch.empty_like(x)
    reps = int(2.5 * 5.0e12 / (4 * n))                      # ~2.5 s of loop at 5 TB/s: long enough for the power reading to settle
    nev = min(reps, 400)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nev)]
    for _ in range(3): y.copy_(x); z.copy_(y)
    torch.cuda.synchronize()
    with Poll([card]) as p:
        t0 = time.time()
        for i in range(reps):
            y.copy_(x)
            if i < nev:
                ev[i][0].record(); z.copy_(y); ev[i][1].record()
            else:
                z.copy_(y)
        torch.cuda.synchronize()
        wall = time.time() - t0
    ms = sorted(e0.elapsed_time(e1) for e0, e1 in ev)[len(ev) // 2]
    print(f"{mb:5d} MB tensors: consumer pass (read + write) {2 * n / ms / 1e9:7.2f} TB/s   ({ms * 1e3:8.1f} us)   loop: {4 * n * reps / wall / 1e12:5.2f} TB/s of tensor bytes at {med([r[0] for r in p.rows[len(p.rows) // 3:]]) / 1e6:6.0f} W", flush=True)
    del x, y, z
