#!/bin/bash
# round 4: final tree check: GPU tests, smoke, short headline
OUT=gpurun_out/r04ab; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4) > $OUT/gpu_tests.txt; cat $OUT/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 600 python bench.py --no-configs --no-cpu-baseline > $OUT/h.json 2> $OUT/h.err; grep "\[bench\]" $OUT/h.err | head -3 | cut -c1-200
timeout 600 python bench.py --config 4 --no-cpu-baseline > $OUT/c4.json 2> $OUT/c4.err; grep "\[bench\] config 4:" $OUT/c4.err | cut -c1-170
