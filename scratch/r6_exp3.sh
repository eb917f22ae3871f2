#!/bin/bash
# round 6, experiment 3: the scan's hit arrays copied to the host beside the next step's counts (kv_hits_lazy) against --eager-hits
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r6_exp3; mkdir -p $OUT
cd $REPO
timeout 900 python3 -m pytest tests/test_gpu_skm.py tests/test_gpu_pipeline.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
L=kevlar_amd/libkvsketch_hip.so
scratch/ab.sh r6_exp3 -- lazy=$L
scratch/ab.sh r6_exp3/eager --eager-hits -- eager=$L
scratch/ab.sh r6_exp3 -- lazy2=$L
scratch/ab.sh r6_exp3/eager --eager-hits -- eager2=$L
