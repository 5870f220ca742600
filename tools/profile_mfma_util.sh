#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 counter pass for MFMA utilisation per kernel of one bench step.
#   tools/profile_mfma_util.sh TAG [extra bench flags]
set -e
tag=${1:-cur}; shift || true
out=gpurun_out/prof_util_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# durations come from a separate kernel-trace pass: under --pmc the dispatch timestamps carry a variable collection overhead (the same
# kernel measured 2.57 and 2.88 ms in two counter passes against 2.42 ms traced), the cycle counters do not
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out/trace_log.txt 2>&1
cp $(ls $out/t/*/*_kernel_stats.csv | head -1) $out/kernel_stats.csv
rm -rf $out/t
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $out/p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $out/log.txt 2>&1
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE; do python3 tools/pmc_summary.py $out/p $c $out/counters.json > /dev/null; done
python3 - $out/kernel_stats.csv $out/counters.json <<'PY'
import csv, json, sys
d = json.load(open(sys.argv[2]))
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"] in d: d[r["Name"]]["duration_ns_traced"] = float(r["AverageNs"])
json.dump(d, open(sys.argv[2], "w"), indent=1, sort_keys=True)
PY
rm -rf $out/p
python3 - $out/counters.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"{'kernel':58s} {'disp':>4s} {'clk GHz*':>8s} {'MFMA busy %':>11s} {'waves parked %':>14s}")
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0) * kv[1].get("dispatches", 0)):
    if v.get("SQ_INSTS_MFMA", 0) <= 0 or "GRBM_GUI_ACTIVE" not in v: continue
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0                       # the counter sums the 8 XCDs
    util = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)   # busy cycles summed over 256 CUs x 4 SIMDs
    parked = v["SQ_WAIT_ANY"] / max(v["SQ_WAVE_CYCLES"], 1.0)
    ns = v.get("duration_ns_traced", 0.0)
    ghz = f"{cyc / ns:8.2f}" if ns > 0 else "       -"                 # shader clock: the kernel's cycles / its traced mean duration
    print(f"{k[:58]:58s} {v['dispatches']:4d} {ghz} {100 * util:10.1f}% {100 * parked:13.1f}%")
PY
