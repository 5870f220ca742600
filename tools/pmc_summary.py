"""Aggregate rocprofv3 --pmc CSV output per kernel: mean counter value per dispatch.
   python tools/pmc_summary.py <dir> COUNTER [out.json]
With out.json the per-kernel means are merged into that file as {kernel: {COUNTER: mean, "dispatches": n}}."""
import csv, glob, json, os, sys, collections
d, ctr = sys.argv[1], sys.argv[2]
out = sys.argv[3] if len(sys.argv) > 3 else None
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: [0, 0.0])
for f in files:
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != ctr:
            continue
        k = r["Kernel_Name"]
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{k[:70]:70s} dispatches {n:5d}  mean {ctr} {v / n:14.1f}  total {v:16.1f}")
if out:
    data = json.load(open(out)) if os.path.exists(out) else {}
    for k, (n, v) in acc.items():
        e = data.setdefault(k, {})
        e[ctr] = v / n
        e["dispatches"] = n
    json.dump(data, open(out, "w"), indent=1, sort_keys=True)
