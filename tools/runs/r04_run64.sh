#!/bin/bash
# round 4: the any-depth tests with a volume at a dword- but not 16-byte-aligned address
OUT=gpurun_out/r04ay; mkdir -p $OUT
(timeout 600 python -m pytest tests -m gpu -q -k "any_depth or label_alignment" 2>&1 | tail -5) > $OUT/t.txt; cat $OUT/t.txt
