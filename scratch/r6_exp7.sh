#!/bin/bash
# round 6, experiment 7: the k = 31 count's table at 4608 / 5120 slots (SKM_TS31) against 4096, config 2
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r6_exp7; mkdir -p $OUT
cd $REPO
timeout 900 python3 -m pytest tests/test_gpu_skm.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
A=scratch/ab/libkv_ts4096.so; L=kevlar_amd/libkvsketch_hip.so; B=scratch/ab/libkv_ts5120.so
scratch/ab.sh r6_exp7/one --count-streams 1 -- ts4096=$A ts4608=$L ts5120=$B
scratch/ab.sh r6_exp7 -- ts4096=$A ts4608=$L ts5120=$B ts4096b=$A ts4608b=$L ts5120b=$B
