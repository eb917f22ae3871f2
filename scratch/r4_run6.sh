#!/bin/bash
# ingest + pipeline + split tests at HEAD (the FASTA .gz paths, partition's new host half against the goldens), then cfg4-band downstream
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r4_run6; mkdir -p $OUT
cd $REPO
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 900 python3 -m pytest tests/test_gpu_ingest.py tests/test_gpu_pipeline.py tests/test_gpu_split_augment.py tests/test_gpu_cfg4_shape.py tests/test_golden_synth.py tests/test_bench_launcher.py -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
timeout 900 python3 bench.py --workload cfg4-band > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
python3 - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_cfg4.json').read().strip().splitlines()[-1])
    print('cfg4-band ms/step', d['ms_per_step'], 'reads/s', d['value'], d['selfcheck'], d['downstream'])
except Exception as e:
    print('cfg4 failed', e)
PY
