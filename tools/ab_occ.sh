#!/bin/bash
# A/B of the halo conv's workgroup shapes inside one gpurun call: vt_set_flag(3, v) = 3: 4 waves x 256 VGPRs, two workgroups per CU (default) for every
# layer; 2: that shape for 128-cout layers only; 1: 8 waves x 128 VGPRs for 128-cout layers; 0: 8 waves x 256 VGPRs, one workgroup per CU, everywhere.
mkdir -p gpurun_out/r3t
for rep in 1 2; do for v in 3 0 1 2; do
python bench.py --no-cpu-baseline --no-also --steps 10 --warmup 3 --flag 3=$v > gpurun_out/r3t/occ_${v}_$rep.json 2>/dev/null || exit 1
python - $v gpurun_out/r3t/occ_${v}_$rep.json <<'PY'
import json, sys
r = json.load(open(sys.argv[2])); pc = r["roofline"]["per_config"]
print(f"flag3={sys.argv[1]} {r['value']:8.2f} img/s {r['ms_per_step']:7.3f} ms | " + "  ".join(f"{k.replace('conv3x3_halo_kernel','halo')}: {v['ms'] / r['steps']:.2f}" for k, v in pc.items() if 'halo' in k), flush=True)
PY
done; done
