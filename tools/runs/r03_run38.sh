#!/bin/bash
OUT=gpurun_out/r03al; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > $OUT/bench_short.json 2> $OUT/err.txt; python -c "
import json; d=json.load(open('$OUT/bench_short.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
