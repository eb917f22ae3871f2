#!/bin/bash
# partition's sorts on the device: tests, then filter / partition on config 4's band output with the profile
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r4_exp12; mkdir -p $OUT
cd $REPO
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 1200 python3 -m pytest tests/test_gpu_pipeline.py tests/test_gpu_cfg4_shape.py -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
KV_E2E_PROFILE=1 timeout 1200 python3 bench.py --workload cfg4-band --steps 1 --warmup 0 > $OUT/bench.json 2> $OUT/bench.err
python3 -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['downstream'])"
grep -A22 "downstream profile. partition" $OUT/bench.err | head -30
