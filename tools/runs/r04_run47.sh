#!/bin/bash
# round 4: kernel trace of the headline step with the lighter instrumentation: what fills the 80 us between two brick launches
OUT=gpurun_out/r04ag; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d "$ROOT/$OUT/trace" -o bench --output-format csv -- python "$ROOT/bench.py" --no-cpu-baseline --no-configs > "$ROOT/$OUT/line.json") > $OUT/trace.log 2>&1
python - <<'PY'
import csv, glob
f=glob.glob('gpurun_out/r04ag/trace/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'siddon_fwd_brick_kernel<true' in r['Kernel_Name']]
k=len(idx)//2
a,b=idx[k],idx[k+1]
t0=int(rows[a]['Start_Timestamp']); prev=None
for r in rows[a:b+1]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print(f"{(s-t0)/1000:9.1f} us  dur {(e-s)/1000:8.1f}  gap {((s-prev)/1000 if prev else 0):6.1f}  {r['Kernel_Name'][:80]}")
    prev=e
print("step period", (int(rows[b]['Start_Timestamp'])-t0)/1000)
PY
