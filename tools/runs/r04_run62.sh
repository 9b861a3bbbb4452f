#!/bin/bash
# round 4: the driver's multi-GPU launch line at N = 1 (torch.distributed.run, RCCL rendezvous on 127.0.0.1)
OUT=gpurun_out/r04aw; mkdir -p $OUT
( time timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 > $OUT/bench_torchrun_n1.json 2> $OUT/bench_torchrun_n1.err ) 2>&1 | grep real; tail -c 600 $OUT/bench_torchrun_n1.json; grep -c . $OUT/bench_torchrun_n1.json; grep -i "error\|traceback" $OUT/bench_torchrun_n1.err | head -5
