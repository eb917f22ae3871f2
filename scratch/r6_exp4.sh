#!/bin/bash
# round 6, experiment 4: the small kernels of the step (loose lists dealt a few records to every wave, k_ab_fill from the list of claimed
# slots, k_tile_scan with its next round requested ahead) -- tests first, then the per-kernel table of one-stream runs
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r6_exp4; mkdir -p $OUT
cd $REPO
timeout 1500 python3 -m pytest tests/test_gpu_skm.py tests/test_gpu_pipeline.py tests/test_gpu_multicase.py tests/test_gpu_longreads.py tests/test_gpu_shard.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
L=kevlar_amd/libkvsketch_hip.so
KV_SKM_VERBOSE=1 timeout 300 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-replay --traffic none --count-streams 1 > $OUT/verbose.json 2> $OUT/verbose.err; grep "kv_skm" $OUT/verbose.err | sort | uniq -c | head -20
scratch/ab.sh r6_exp4/one --count-streams 1 -- new=$L
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r6_exp4/one/new.json').read().strip().splitlines()[-1])
print({k: round(v, 3) for k, v in d['roofline']['kernels_ms_per_step'].items()})
PY
scratch/ab.sh r6_exp4 -- new=$L new2=$L
