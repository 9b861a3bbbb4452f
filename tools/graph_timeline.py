"""Timeline of one registration iteration inside its HIP graph (BASELINE config 4): from a rocprofv3 kernel trace of
`python bench.py --config 4 --no-cpu-baseline`, the kernels of the median replay with their durations and the gaps
between them (development tool).
Usage (on the GPU box, from the repo root):
    cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace -d /tmp/g4 -o x --output-format csv -- python $REPO/bench.py --config 4 --no-cpu-baseline
    python tools/graph_timeline.py /tmp/g4"""
import csv
import glob
import os
import sys

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# an iteration starts with the fused pose -> rays kernel
starts = [i for i, r in enumerate(rows) if "pose_raygen_fwd_kernel" in r[2]]
its = [rows[a:b] for a, b in zip(starts, starts[1:]) if b - a < 40]
its = [it for it in its if any("siddon_fwd_brick_kernel" in k[2] or "siddon_few" in k[2] for k in it)]
if not its:
    sys.exit("no iterations found")
n = max(set(len(it) for it in its), key=[len(it) for it in its].count)
its = [it for it in its if len(it) == n][len(its) // 3:]
print(f"{len(its)} replays of {n} kernels; median durations / gaps in us")
import statistics as st
span = st.median((it[-1][1] - it[0][0]) / 1e3 for it in its)
period = st.median((b[0][0] - a[0][0]) / 1e3 for a, b in zip(its, its[1:]))
tot_k = 0.0
for k in range(n):
    dur = st.median((it[k][1] - it[k][0]) / 1e3 for it in its)
    gap = st.median((it[k][0] - it[k - 1][1]) / 1e3 for it in its) if k else 0.0
    tot_k += dur
    print(f"  gap {gap:6.2f}  {dur:8.2f}  {its[0][k][2][:110]}")
print(f"kernels {tot_k:.1f} us, first start -> last end {span:.1f} us, start -> next start {period:.1f} us")
