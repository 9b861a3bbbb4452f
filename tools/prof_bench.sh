#!/bin/bash
# rocprofv3 evidence for the bench command itself: kernel-trace stats of `bench.py` plus
# separate PMC passes (HBM traffic) of the dominant kernel on the bench workload.
# usage: tools/prof_bench.sh <outdir>
OUT=$1; mkdir -p "$OUT"; export TMPDIR=/tmp; ROOT=$(pwd)
# (the bench's own default step count: the same warm clocks as the HIP-event figures of the line)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/bench_trace" -o bench --output-format csv -- python "$ROOT/bench.py" --no-cpu-baseline --no-configs > "$ROOT/$OUT/bench_line_under_trace.json") > "$OUT/bench_trace.log" 2>&1
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  (cd /tmp && timeout 90 rocprofv3 --pmc $c -d "$ROOT/$OUT/pmc_$n" -o pmc --output-format csv -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-configs) > "$OUT/pmc_$n.log" 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(out, "bench_trace", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
print("## rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-configs (top kernels)")
for r in rows[:12]:
    print(f"{float(r['AverageNs'])/1e6:9.4f} ms avg  x{r['Calls']:>5}  {float(r['Percentage']):6.2f}%  {r['Name'][:110]}")
cnt = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
import json
traffic = {"_comment": "HBM traffic per launch of the brick kernels on bench.py's workload (512^3, 256^2, 32 poses), "
           "from separate rocprofv3 --pmc passes of `python bench.py --steps 3 --warmup 1 --no-cpu-baseline "
           "--no-configs` (tools/prof_bench.sh).  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE counts "
           "half of wide coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section)."}
# (the bench line's forward_f32 leg runs the 32^3 fp32 configuration of the same kernel template)
for key, pat in (("forward_record", "siddon_fwd_brick_kernel<true, (anonymous namespace)::FwdCfg<32, 32, 64"),
                 ("forward", "siddon_fwd_brick_kernel<false, (anonymous namespace)::FwdCfg<32, 32, 64"),
                 ("forward_f32", "siddon_fwd_brick_kernel<false, (anonymous namespace)::FwdCfg<32, 32, 32")):
    for k, d in cnt.items():
        if pat in k and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            f, w = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]), sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
            ent = {"kernel": k[:100], "fetch_size_kb": f, "write_size_kb": w,
                   "hbm_bytes_per_launch": int((2 * f + w) * 1024), "launches": len(d["FETCH_SIZE"])}
            for c, name in (("TCC_ATOMIC_sum", "tcc_atomic_requests"), ("TCC_HIT_sum", "tcc_hit"), ("TCC_REQ_sum", "tcc_req")):
                if c in d:
                    ent[name] = sum(d[c]) / len(d[c])
            traffic[key] = ent
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print("## PMC per dispatch (separate passes), kernels with 'siddon' in the name")
for k, d in cnt.items():
    if "siddon" not in k: continue
    print(k[:120])
    for c, v in sorted(d.items()):
        print(f"   {c:18s} mean {sum(v)/len(v):.6g}  (n={len(v)})")
PY
