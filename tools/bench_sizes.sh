mkdir -p gpurun_out/r3w
for args in "--batch 1" "--batch 1 --fp8" "--batch 4" "--batch 4 --fp8" "--batch 32" "--batch 32 --fp8" "--encode-only --batch 8" "--bucketed" "--bucketed --fp8" "--batch 1 --height 512 --width 512"; do
  n=$(echo $args | tr -d ' -')
  python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 5 $args > gpurun_out/r3w/$n.json 2>gpurun_out/r3w/$n.err || { echo "FAILED $args"; tail -3 gpurun_out/r3w/$n.err; exit 1; }
  python - "$args" gpurun_out/r3w/$n.json <<'PY'
import json, sys
r = json.load(open(sys.argv[2]))
print(f"{sys.argv[1]:40s} {r['value']:8.2f} img/s {r['ms_per_step']:8.3f} ms/step  power {r['power']['socket_w_median'] if r.get('power') else None} W", flush=True)
PY
done
