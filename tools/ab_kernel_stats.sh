#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel statistics of a short bench for each library build; prints the rows matching PATTERN.
#   tools/ab_kernel_stats.sh PATTERN lib1.so lib2.so ... [-- extra bench flags]
pat=$1; shift
libs=(); extra=()
while [ $# -gt 0 ]; do if [ "$1" == "--" ]; then shift; extra=("$@"); break; fi; libs+=("$1"); shift; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for lib in "${libs[@]}"; do
  name=$(basename $lib .so); out=gpurun_out/abks_$name
  rm -rf $out; mkdir -p $out
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-also --lib $lib "${extra[@]}" > $out/log.txt 2>&1 || { echo FAILED $lib; tail -3 $out/log.txt; exit 1; }
  f=$(ls $out/t/*/*_kernel_stats.csv | head -1)
  python3 - "$name" "$f" "$pat" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[2])):
    if sys.argv[3] in r["Name"]:
        print(f"{sys.argv[1]:24s} {r['Name'][:50]:50s} calls {r['Calls']:>4s} avg {float(r['AverageNs']) / 1e3:9.1f} us  min {int(r['MinNs']) / 1e3:9.1f}", flush=True)
PY
  rm -rf $out/t
done; done
