#!/bin/bash
# round 6, experiment 2: the drain's murmurs from the product tables + 16-bit occurrence counters (SKM_PL) against the build without
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r6_exp2; mkdir -p $OUT
cd $REPO
timeout 900 python3 -m pytest tests/test_gpu_skm.py tests/test_gpu_sketch.py tests/test_gpu_kmer2bit.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
N=scratch/ab/libkv_nopl.so; L=kevlar_amd/libkvsketch_hip.so
scratch/ab.sh r6_exp2/one --count-streams 1 -- nopl=$N pl=$L
scratch/ab.sh r6_exp2 -- nopl=$N pl=$L nopl2=$N pl2=$L
scratch/ab.sh r6_exp2/cfg5 --workload cfg5 --count-streams 1 -- nopl=$N pl=$L
