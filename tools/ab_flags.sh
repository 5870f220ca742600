#!/bin/bash
# Run ON THE GPU BOX: interleaved A/B of bench.py variants on one box.   tools/ab_flags.sh OUT REPS "<args A>" "<args B>" ...
out=$1; reps=$2; shift 2
: > $out
for r in $(seq 1 $reps); do
  for v in "$@"; do
    line=$(python3 bench.py --no-cpu-baseline --no-also --steps 15 --warmup 4 $v 2>/dev/null | tail -1)
    python3 - "$r" "$v" "$line" >> $out <<'PY'
import json, sys
r, v, line = sys.argv[1:4]
d = json.loads(line)
pc = d["roofline"]["per_config"]
print(f"rep {r} [{v or 'default':28s}] {d['value']:7.2f} images/s  {d['ms_per_step_without_events']:7.3f} ms/step  " + "  ".join(f"{k.split('_kernel')[0]}{k.split('_kernel')[1][:12]}: {x['ms'] / d['steps']:.3f}" for k, x in pc.items()))
PY
  done
done
cat $out
