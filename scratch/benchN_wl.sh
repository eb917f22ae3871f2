#!/bin/bash
# bash scratch/benchN_wl.sh WORKLOAD N [extra bench args]: N ranks over gloo on the one GPU for another workload
wl=$1; n=$2; shift 2
mkdir -p gpurun_out
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2971$n bench.py --workload $wl --gpus $n --steps 1 --warmup 1 --backend gloo --no-cpu-baseline --no-e2e "$@" > gpurun_out/bench_gloo_${wl}_$n.json 2> gpurun_out/bench_gloo_${wl}_$n.err
echo "$wl N=$n $* rc=$?"; grep -o "\"selfcheck[^}]*}" gpurun_out/bench_gloo_${wl}_$n.json | cut -c1-500; grep "Error" gpurun_out/bench_gloo_${wl}_$n.err | grep rank0 | head -3
