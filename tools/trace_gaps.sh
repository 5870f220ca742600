#!/bin/bash
# Run ON THE GPU BOX: is a small shape launch-bound?  One traced bench step: wall time from the first kernel's start to the last kernel's end, the sum of
# the kernel durations, and the idle time between consecutive kernels (gpurun_out/trace_gaps.txt).   tools/trace_gaps.sh --batch 1 --height 512 --width 512
set -e
out=gpurun_out/trace_gaps
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-also "$@" > $out/log.txt 2>&1
python3 - "$@" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/trace_gaps/t/*/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# steps are separated by the largest gaps (host synchronisation between timed regions); take the last step = launches after the last big gap
n = len(rows)
per = None
for k in range(n - 1, 0, -1):
    if "conv_in" in rows[k][2]:
        first = k; break
last = rows[first:]
# (bench.py's parity leg follows the timed steps: cut at the one large gap, if any)
g0 = [(last[i + 1][0] - last[i][1]) for i in range(len(last) - 1)]
if g0 and max(g0) > 1e6:
    last = last[:g0.index(max(g0)) + 1]
wall = (last[-1][1] - last[0][0]) / 1e3
busy = sum(e - s for s, e, _ in last) / 1e3
gaps = [(last[i + 1][0] - last[i][1]) / 1e3 for i in range(len(last) - 1)]
with open("gpurun_out/trace_gaps.txt", "a") as o:
    o.write(f"{' '.join(sys.argv[1:]) or '(default)'}: {len(last)} launches in the last step; first start -> last end {wall:.1f} us; kernels {busy:.1f} us; idle between kernels {sum(g for g in gaps if g > 0):.1f} us "
            f"(median gap {sorted(gaps)[len(gaps) // 2]:.2f} us, max {max(gaps):.1f} us)\n")
print(open("gpurun_out/trace_gaps.txt").read())
PY
rm -rf $out/t
