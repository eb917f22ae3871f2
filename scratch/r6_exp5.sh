#!/bin/bash
# round 6, experiment 5: k_skm_count combining PAIRS of consecutive k-mers (SKM_PAIRS) against the build without
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r6_exp5; mkdir -p $OUT
cd $REPO
timeout 1500 python3 -m pytest tests/test_gpu_skm.py tests/test_gpu_sketch.py tests/test_gpu_pipeline.py tests/test_gpu_multicase.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
N=scratch/ab/libkv_nopairs.so; L=kevlar_amd/libkvsketch_hip.so
KV_SKM_VERBOSE=1 timeout 300 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-replay --traffic none --count-streams 1 2>&1 >/dev/null | grep "kv_skm\] batch" | sort | uniq -c | head
scratch/ab.sh r6_exp5/one --count-streams 1 -- nopairs=$N pairs=$L
scratch/ab.sh r6_exp5 -- nopairs=$N pairs=$L nopairs2=$N pairs2=$L
