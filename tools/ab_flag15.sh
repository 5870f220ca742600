mkdir -p gpurun_out/r3v
for rep in 1 2; do for v in 1 0; do
python bench.py --no-cpu-baseline --fp8 --steps 10 --warmup 3 --flag 15=$v > gpurun_out/r3v/p15_${v}_$rep.json 2>/dev/null || exit 1
python - $v gpurun_out/r3v/p15_${v}_$rep.json <<'PY'
import json, sys
r = json.load(open(sys.argv[2])); pc = r["roofline"]["per_config"]
print(f"flag15={sys.argv[1]} {r['value']:8.2f} img/s {r['ms_per_step']:7.3f} ms | " + "  ".join(f"{k}: {v['ms'] / r['steps']:.2f}" for k, v in pc.items() if 'halo' not in k) + f" | dlogit {r['config'].get('max_abs_dlogit_vs_oracle')}", flush=True)
PY
done; done
