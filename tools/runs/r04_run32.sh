#!/bin/bash
# round 4: look-ahead in the product build: GPU tests, bench headline + config 4
OUT=gpurun_out/r04u; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > $OUT/tests.txt; cat $OUT/tests.txt
(timeout 600 python bench.py --no-configs --no-cpu-baseline > $OUT/bench_headline.json) 2> $OUT/bench_headline.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r04u/bench_headline.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline']['forward']['frac'], d['roofline']['forward']['kernel_ms'])
PY
(timeout 600 python bench.py --config 4 --no-cpu-baseline > $OUT/bench_config_4.json) 2> $OUT/bench_config_4.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r04u/bench_config_4.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['kernel_ms'])
PY
