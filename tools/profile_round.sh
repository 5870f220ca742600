#!/bin/bash
# Run ON THE GPU BOX (gpurun): kernel-trace statistics of the default bench, then HBM traffic counters in two separate
# --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950).  Outputs under gpurun_out/prof_$1/.
#   tools/profile_round.sh TAG [extra bench.py flags, e.g. --fp8]
set -e
tag=${1:-cur}
shift || true
extra="$@"
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-also $extra > $out/trace.log 2>&1
cp $(ls $out/trace/*/*_kernel_stats.csv | head -1) $out/kernel_stats.csv
grep '^{' $out/trace.log > $out/bench_under_trace.json || true
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-also $extra > $out/pmc_$c.log 2>&1
  python3 tools/pmc_summary.py $out/pmc_$c $c $out/pmc_traffic.json > $out/pmc_$c.txt
done
rm -rf $out/trace $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
ls -la $out
