#!/bin/bash
# round 4: the forward-only walk with accumulated chord-relative alphas (16 instead of 19 instructions per step)
OUT=gpurun_out/r04d; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $OUT/gpu_tests.txt; tail -25 $OUT/gpu_tests.txt
timeout 600 python bench.py > $OUT/bench_headline.json 2> $OUT/bench_headline.err; grep "\[bench\]" $OUT/bench_headline.err
python - <<'PY'
import json
r=json.load(open("gpurun_out/r04d/bench_headline.json"))
print("value", r["value"], "ms", r["ms_per_step"], "frac", r["roofline"]["frac"], "fwd", r["roofline"]["forward"]["frac"], r["roofline"]["forward"]["kernel_ms"])
print("parity", {k:v for k,v in r["parity"].items() if isinstance(v,(int,float))})
print("forward_sweep", {k:v for k,v in r["roofline"].get("forward_sweep",{}).items() if k in("kernel_ms","frac","parity")})
PY
# the record kernel: grouping of 8 / aligned rows on and off (tools build)
timeout 600 python tools/brick_bench.py --cases pert32aux,pert32 --dbg 0,8,1024,1032 2>&1 | grep -v amdgpu.ids > $OUT/record_grouping.txt; cat $OUT/record_grouping.txt
