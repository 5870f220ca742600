#!/bin/bash
# Run ON THE GPU BOX: per-dispatch kernel durations of one bench step, in launch order (gpurun_out/trace_step.txt).
set -e
out=gpurun_out/trace_step
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $out/log.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace_step/t/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // 2
with open("gpurun_out/trace_step.txt", "w") as o:
    for r in rows[n:]:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
        o.write(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:10.1f} us  grid {r['Grid_Size_X']:>10}  {name}\n")
PY
rm -rf $out/t
