#!/bin/bash
mkdir -p gpurun_out/hits
export BENCH_DUMP_HITS=$PWD/gpurun_out/hits
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29614 bench.py --gpus 4 --steps 1 --warmup 0 --backend gloo --multi exchange --exchange-items plain --no-cpu-baseline --no-e2e > /dev/null 2> gpurun_out/dbg4.err
grep AssertionError gpurun_out/dbg4.err | head -1
timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e 2>/dev/null | grep -o "\"banded_checksums[^}]*}"
ls -la gpurun_out/hits
