#!/bin/bash
# round 4, run 4: cfg4-band downstream with the file-descriptor formatter (+ cProfile of filter / partition), kernel stats of the step,
# the FASTA .gz test, the launcher test (projection block)
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r4_run4; mkdir -p $OUT
cd $REPO
timeout 600 python3 -m pytest tests/test_gpu_ingest.py tests/test_bench_launcher.py tests/test_gpu_split_augment.py -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
KV_E2E_PROFILE=1 timeout 1200 python3 bench.py --workload cfg4-band > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
grep -A28 "downstream profile" $OUT/bench_cfg4.err | cut -c1-160 | head -90
python3 - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_cfg4.json').read().strip().splitlines()[-1])
    print('cfg4-band ms/step', d['ms_per_step'], 'reads/s', d['value'], d['selfcheck'], d['downstream'])
except Exception as e:
    print('cfg4 failed', e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg4 -- python3 $REPO/bench.py --workload cfg4-band --steps 1 --warmup 1 --no-downstream --count-streams 1 > $OUT/bench_cfg4_under_rocprof.json 2> $OUT/trace_cfg4.err
python3 - <<PY
import csv, glob
fs = glob.glob('$OUT/trace_cfg4/**/*kernel_stats.csv', recursive=True)
if fs:
    rows = list(csv.reader(open(fs[0])))
    print(rows[0])
    for r in rows[1:12]: print([c[:60] for c in r])
PY
