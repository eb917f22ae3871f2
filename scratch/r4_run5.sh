#!/bin/bash
# clean re-run of the whole GPU suite at 54eeb32 + cfg4-band with its downstream stages
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r4_run5; mkdir -p $OUT
cd $REPO
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 1300 python3 -m pytest tests -m gpu -q --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -16 $OUT/pytest.log
timeout 900 python3 bench.py --workload cfg4-band > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
python3 - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_cfg4.json').read().strip().splitlines()[-1])
    print('cfg4-band ms/step', d['ms_per_step'], 'reads/s', d['value'], d['selfcheck'], d['downstream'])
except Exception as e:
    print('cfg4 failed', e)
PY
