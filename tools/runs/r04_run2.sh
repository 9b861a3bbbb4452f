#!/bin/bash
# round 4: the default bench line with configs / sweep / forward_sweep; the launch-state stress test
OUT=gpurun_out/r04b; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_brick_storage.py -x -q -k "streams or empty" 2>&1 | tail -5 > $OUT/stress.txt; cat $OUT/stress.txt
( time timeout 900 python bench.py > $OUT/bench_headline.json 2> $OUT/bench_headline.err ) 2>&1 | tail -4
grep "\[bench\]" $OUT/bench_headline.err; tail -5 $OUT/bench_headline.err
python - <<'PY'
import json
r=json.load(open("gpurun_out/r04b/bench_headline.json"))
print("value", r["value"], "ms", r["ms_per_step"], "frac", r["roofline"]["frac"], "fwd", r["roofline"]["forward"]["frac"])
print("forward_sweep", {k:v for k,v in r["roofline"].get("forward_sweep",{}).items() if k in("kernel_ms","frac","parity")})
print("sweep", r.get("sweep"))
for k,c in r.get("configs",{}).items():
    print(k, c["value"], c["unit"], c["ms_per_step"], c["dominant_kernel"], c["parity"], c["wall_s"])
print(r["config"])
PY
