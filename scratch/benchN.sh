#!/bin/bash
# N ranks sharing the one GPU over gloo, the layout bench.py would pick by itself at that N (auto): validates the multi-GPU
# code path end to end (ranks_seen, k-mer totals, merged hits == band-by-band replay): bash scratch/benchN.sh 4 8
mkdir -p gpurun_out
for n in "$@"; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 1 --warmup 1 --backend gloo --no-cpu-baseline --no-e2e > gpurun_out/bench_gloo_$n.json 2> gpurun_out/bench_gloo_$n.err
  echo "N=$n rc=$?"; grep -o "\"parallelism[^}]*\|\"selfcheck[^}]*}" gpurun_out/bench_gloo_$n.json | cut -c1-700; tail -2 gpurun_out/bench_gloo_$n.err | cut -c1-300
done
