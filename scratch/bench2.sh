#!/bin/bash
# two ranks sharing the one GPU over gloo: the exchange layout with both item kinds, then the banded layout
mkdir -p gpurun_out
for items in distinct plain; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --multi exchange --exchange-items $items --no-cpu-baseline --no-e2e > gpurun_out/bench2_$items.json 2> gpurun_out/bench2_$items.err
  tail -c 2500 gpurun_out/bench2_$items.json; tail -3 gpurun_out/bench2_$items.err
done
