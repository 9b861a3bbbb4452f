#!/bin/bash
OUT=gpurun_out/r03f; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
python bench.py > $OUT/bench_headline.json 2> $OUT/bench_headline.err; tail -4 $OUT/bench_headline.err; cat $OUT/bench_headline.json
python bench.py --storage f32 --no-cpu-baseline > $OUT/bench_headline_f32.json 2> $OUT/bench_headline_f32.err; tail -3 $OUT/bench_headline_f32.err
