#!/bin/bash
# Run ON THE GPU BOX: tools/bin/fetch_probe plain (timing) and under rocprofv3 --pmc FETCH_SIZE (what the counter reports per access shape).
set -e
out=gpurun_out/fetch_probe
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 120 tools/bin/fetch_probe 2 > $out/timing.log 2>&1
cat $out/timing.log
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/p -- tools/bin/fetch_probe 2 > $out/pmc_run.log 2>&1
python3 - $out/p <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == "FETCH_SIZE":
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k[:60]:60s} dispatches {len(v):2d}  FETCH_SIZE per dispatch (KB, raw): " + " ".join(f"{x:12.0f}" for x in v))
PY
rm -rf $out/p
