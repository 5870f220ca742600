"""What does the socket draw, and what clock does the SMU report, while the step runs?  Polls rocm-smi / amd-smi and the hwmon files of every amdgpu
card that is visible while `bench.py` (child process) runs its timed steps.     python tools/power_poll.py [--fp8]
Read beside DESIGN §4.12: the in-kernel clock says the kernels run at 1.55-1.9 of 2.4 GHz; this says whether the reason is the power cap."""
import glob, os, subprocess, sys, time, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def read(p):
    try:
        return open(p).read().strip()
    except Exception:
        return None
hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
def smi(args):
    try:
        return subprocess.run(args, capture_output=True, text=True, timeout=20).stdout
    except Exception as e:
        return f"({args[0]}: {e})"
print(smi(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--showtemp"]), flush=True)
extra = sys.argv[1:]
child = subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-also", "--steps", "150", "--warmup", "5"] + extra,
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
rows = []
t0 = time.time()
while child.poll() is None:
    row = {"t": round(time.time() - t0, 2)}
    for h in hw:
        card = h.split("/")[4]
        for f in ("power1_average", "power1_input", "freq1_input", "freq2_input", "temp1_input", "temp2_input", "temp3_input", "power1_cap"):
            v = read(os.path.join(h, f))
            if v is not None:
                row[f"{card}.{f}"] = int(v)
    rows.append(row)
    time.sleep(0.25)
line = child.stdout.read().strip().splitlines()[-1]
r = json.loads(line)
print(f"bench: {r['value']:.1f} img/s, {r['ms_per_step']:.2f} ms/step, dtype {r['dtype']}", flush=True)
busy = [row for row in rows if row["t"] > 0.6 * rows[-1]["t"]]         # the last 40 % of the run: inside the timed steps
# the hwmon tree shows every card of the host; ours is the one whose power rose with the bench
def rise(card):
    v = sorted(row.get(f"{card}.power1_input", 0) for row in busy)
    return (v[len(v) // 2] if v else 0) - (rows[0].get(f"{card}.power1_input") or 0)
mine = max({h.split("/")[4] for h in hw}, key=rise)
keys = sorted({k for row in rows for k in row if k.startswith(mine + ".")})
for k in keys:
    vals = [row[k] for row in busy if k in row]
    if vals and max(vals) > 0:
        print(f"{k:34s} during the timed steps: min {min(vals):>12d}  median {sorted(vals)[len(vals)//2]:>12d}  max {max(vals):>12d}   (idle at start: {rows[0].get(k)})")
print(smi(["rocm-smi", "--showpower", "--showclocks"]), flush=True)
