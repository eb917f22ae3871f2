#!/bin/bash
# round 6, experiment 6: the k = 51 count's table at 2560 slots (SKM_TS51) against 2048, config 5
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r6_exp6; mkdir -p $OUT
cd $REPO
timeout 900 python3 -m pytest tests/test_gpu_skm.py tests/test_gpu_longreads.py -m gpu -q -x -k "51 or k_equals or long" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
N=scratch/ab/libkv_ts2048.so; L=kevlar_amd/libkvsketch_hip.so
scratch/ab.sh r6_exp6/one --workload cfg5 --count-streams 1 -- ts2048=$N ts2560=$L
scratch/ab.sh r6_exp6 --workload cfg5 -- ts2048=$N ts2560=$L ts2048b=$N ts2560b=$L
