#!/bin/bash
# A/B of library builds inside ONE gpurun call (box-to-box spread is ~2 %): tools/ab.sh TAG lib1.so lib2.so ... [-- extra bench flags]
# Runs bench.py for each library twice, interleaved, and prints images/s + per-kernel ms from the bench line.
tag=$1; shift
libs=(); extra=()
while [ $# -gt 0 ]; do if [ "$1" == "--" ]; then shift; extra=("$@"); break; fi; libs+=("$1"); shift; done
mkdir -p gpurun_out
for rep in 1 2; do
  for lib in "${libs[@]}"; do
    name=$(basename "$lib" .so)
    python bench.py --no-cpu-baseline --steps 10 --warmup 3 --lib "$lib" "${extra[@]}" > gpurun_out/ab_${tag}_${name}_$rep.json 2> gpurun_out/ab_${tag}_${name}_$rep.err || { echo "FAILED $lib"; tail -5 gpurun_out/ab_${tag}_${name}_$rep.err; exit 1; }
    python - "$name" gpurun_out/ab_${tag}_${name}_$rep.json <<'PY'
import json, sys
r = json.load(open(sys.argv[2]))
pc = r["roofline"]["per_config"]
print(f"{sys.argv[1]:28s} {r['value']:8.2f} img/s  {r['ms_per_step']:7.3f} ms | " + "  ".join(f"{k.replace('conv3x3_halo_kernel','halo').replace('conv_gemm_kernel','gemm')}: {v['ms'] / r['steps']:.2f}" for k, v in pc.items()) + f" | gn {r['hbm_pass']['avg_launch_ms'] * r['hbm_pass']['launches'] / r['steps']:.2f}", flush=True)
PY
  done
done
