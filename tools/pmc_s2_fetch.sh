#!/bin/bash
# Run ON THE GPU BOX: FETCH_SIZE / WRITE_SIZE of the stride-2 phase-plane kernels with the NHWC (flag 19 = 0) and the chunk-planar (default) input.
#   tools/pmc_s2_fetch.sh [--fp8]
set -e
out=gpurun_out/pmc_s2
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 0 1; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $out/p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-also --flag 19=$v "$@" > $out/log.txt 2>&1
    python3 tools/pmc_summary.py $out/p $ctr $out/flag19_$v.json | grep -i "s2_halo" || true
    rm -rf $out/p
  done
done
python3 - $out <<'PY'
import json, sys
for v in (0, 1):
    d = json.load(open(f"{sys.argv[1]}/flag19_{v}.json"))
    for k, c in d.items():
        if "s2_halo" in k and "FETCH_SIZE" in c:
            print(f"flag 19 = {v}: {k[:60]:60s} {c['dispatches']} dispatches: fetched {2 * c['FETCH_SIZE'] * 1024 / 1e9:.2f} GB (2 x FETCH_SIZE), written {c.get('WRITE_SIZE', 0) * 1024 / 1e9:.2f} GB per launch (mean over the three layers)")
PY
