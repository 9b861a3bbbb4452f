"""What binds the brick kernels: VALU wave-instructions per launch (rocprofv3 --pmc, separate
passes through tools/prof_cases.py), the walk's wave-steps (profile build, tools/brick_profile.py
counters) and, for the LDS-accumulating volume-gradient kernel, the LDS pipe's busy cycles ->
profiles/rNN/issue_bound.json, which bench.py reads into `roofline...issue_bound`.
    python tools/issue_bound.py profiles/r04          (on the MI355X box, from the repo root)
Issue time per wave-instruction and SIMD: 1.05 ns, the measured fast-class rate
(profiles/r02/ubench_valu_rates.txt: v_add/sub/mul/fma/mov/and/or at 4 waves per SIMD); the
slow-class opcodes (1.75 ns) make the true issue time longer, so issue_ms is a lower bound."""
import csv
import glob
import json
import os
import re
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/issue_bound"
scratch = os.path.join(ROOT, "gpurun_out", "issue_bound_scratch")
os.makedirs(scratch, exist_ok=True)
os.makedirs(os.path.join(ROOT, out_dir), exist_ok=True)
NS_PER_INST, SIMDS, CUS, CLOCK_GHZ = 1.05, 1024, 256, 2.4

CASES = {  # key -> (prof_cases args, substring of the kernel name, profile-build case or None, insts per step)
    "forward": (["--kernel", "brick", "--case", "pert32", "--reps", "12"], "siddon_fwd_brick_kernel<false", "pert32", 16),
    "forward_record": (["--kernel", "brick", "--case", "pert32", "--aux", "1", "--reps", "12"],
                       "siddon_fwd_brick_kernel<true", "pert32aux", 27),
    "forward_sweep": (["--kernel", "brick", "--case", "pert512", "--reps", "4"], "siddon_fwd_brick_kernel<false", "pert512", 16),
    "trilinear_volume_gradient": (["--kernel", "trivol", "--case", "pert1", "--det", "512", "--reps", "12"],
                                  "siddon_brick_kernel<4>", None, None),
    "trilinear_forward": (["--kernel", "trifwd", "--case", "pert1", "--det", "512", "--reps", "12"],
                          "siddon_brick_kernel<3>", None, None),
}
PMC = [["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"],
       ["SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_SALU"]]


def rocprof(tag, rp_args, pc_args):
    d = os.path.join(scratch, tag)
    cmd = ["rocprofv3", *rp_args, "-d", d, "-o", "x", "--output-format", "csv", "--", sys.executable,
           os.path.join(ROOT, "tools", "prof_cases.py"), *pc_args]
    env = dict(os.environ, TMPDIR="/tmp")
    with open(os.path.join(scratch, tag + ".log"), "w") as log:
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT, timeout=240, check=False)
    return d


result = {"_comment": __doc__.strip().splitlines()[0] + " ... (tools/issue_bound.py; one MI355X)",
          "ns_per_valu_wave_inst_per_simd": NS_PER_INST, "simds": SIMDS, "kernels": {}}
for key, (pc_args, kname, prof_case, ips) in CASES.items():
    ent = {}
    d = rocprof(key + "_trace", ["--kernel-trace"], pc_args)
    durs = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kname in r["Kernel_Name"]:
                durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    if durs:
        warm = durs[len(durs) // 3:]  # (the first launches run at idle clocks)
        ent["kernel_ms"] = sum(warm) / len(warm)
    for i, counters in enumerate(PMC):
        d = rocprof(f"{key}_pmc{i}", ["--pmc", *counters], pc_args)
        acc = defaultdict(list)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if kname in r["Kernel_Name"]:
                    acc[(r["Counter_Name"], r["Dispatch_Id"])].append(float(r["Counter_Value"]))
        per = defaultdict(list)
        for (c, _), v in acc.items():
            per[c].append(sum(v))
        for c, v in per.items():
            ent[c] = sum(v) / len(v)
    if "SQ_INSTS_VALU" in ent:
        ent["valu_wave_insts"] = ent["SQ_INSTS_VALU"]
    if "SQ_LDS_IDX_ACTIVE" in ent:
        ent["lds_busy_ms"] = ent["SQ_LDS_IDX_ACTIVE"] / CUS / (CLOCK_GHZ * 1e6)
        ent["lds_wave_insts"] = ent.get("SQ_INSTS_LDS")
    if prof_case:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "brick_profile.py"), "--cases", prof_case,
                            "--storage", "q16p"], capture_output=True, text=True, timeout=600, cwd=ROOT)
        m = re.search(r"#wave-steps\s+(\d+)", p.stdout)
        h = re.search(r"#hits\s+(\d+)", p.stdout)
        bt = re.search(r"#batches\s+(\d+)", p.stdout)
        if m:
            ent["walk_wave_steps"] = int(m.group(1))
            ent["hits"] = int(h.group(1)) if h else None
            ent["batches"] = int(bt.group(1)) if bt else None
            ent["insts_per_step"] = ips
        with open(os.path.join(ROOT, out_dir, f"phase_profile_{key}.txt"), "w") as f:
            f.write(p.stdout[-6000:])
    result["kernels"][key] = ent
    print(key, json.dumps(ent), flush=True)
with open(os.path.join(ROOT, out_dir, "issue_bound.json"), "w") as f:
    json.dump(result, f, indent=1)
print("wrote", os.path.join(out_dir, "issue_bound.json"))
