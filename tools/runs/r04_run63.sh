#!/bin/bash
# round 4: kernel durations under rocprofv3 (no launch gaps) at one pose: general kernel, configurable fp32 32^3, packed 16-bit bricks
OUT=gpurun_out/r04ax; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/trace" -o t --output-format csv -- python "$ROOT/tools/brick_bench.py" --variants=-1,0 --storage f32 --cases pert1,pert1aux,pert4 ) > $OUT/trace.log 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/trace2" -o t --output-format csv -- python "$ROOT/tools/brick_bench.py" --variants=-2 --storage q16p --cases pert1,pert1aux,pert4 ) > $OUT/trace2.log 2>&1
grep "ms (best" $OUT/trace.log $OUT/trace2.log | cut -c1-190
python - <<'PY'
import csv, glob
for d in ("trace", "trace2"):
    for f in glob.glob(f"gpurun_out/r04ax/{d}/**/*kernel_stats.csv", recursive=True):
        for r in list(csv.DictReader(open(f)))[:8]:
            print(d, f"{float(r['AverageNs'])/1e3:9.1f} us avg  min {float(r['MinNs'])/1e3:8.1f}  x{r['Calls']:>5}  {r['Name'][:100]}")
PY
