"""Summarise a tools/pmc_run.sh output directory: per kernel, average duration (kernel
trace) and per-dispatch mean of every PMC counter collected (separate passes)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
dur = defaultdict(list)
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
cnt = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
meta = {}
for f in glob.glob(os.path.join(out, "pmc_valu", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        meta[r["Kernel_Name"]] = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count",
                                                        "LDS_Block_Size", "Scratch_Size",
                                                        "Workgroup_Size", "Grid_Size")}
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    short = k[:110]
    d = dur[k]
    print(f"## {short}\n  launches {len(d)}  avg {sum(d) / len(d):.4f} ms  min {min(d):.4f}  total {sum(d):.3f} ms")
    if k in meta:
        print("  " + "  ".join(f"{a}={b}" for a, b in meta[k].items()))
    for c, v in sorted(cnt.get(k, {}).items()):
        # rocprofv3 emits one row per dispatch (already summed over dimensions) or several; average per dispatch
        n = max(1, len(d)) if len(v) % max(1, len(d)) == 0 else len(v)
        print(f"  {c:28s} {sum(v) / max(1, len(d)):.6g}  (rows {len(v)})")
