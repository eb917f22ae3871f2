#!/bin/bash
# round 4, run 3: split-layout fine items -- parity first, then the bench lines and the HBM bytes; cfg4-band with the scan that remembers
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r4_run3; mkdir -p $OUT
cd $REPO
timeout 1200 python3 -m pytest tests/test_gpu_binned.py tests/test_gpu_skm.py tests/test_gpu_sketch.py tests/test_gpu_pipeline.py tests/test_gpu_shard.py tests/test_gpu_fullsize.py tests/test_gpu_cfg4_shape.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -8 $OUT/pytest.log
L=kevlar_amd/libkvsketch_hip.so
bash scratch/ab.sh r4_run3/cfg2_1s --count-streams 1 -- new=$L new_b=$L
bash scratch/ab.sh r4_run3/cfg2_3s -- new=$L new_b=$L
bash scratch/ab.sh r4_run3/cfg5_3s --workload cfg5 -- new=$L
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-replay --count-streams 1 --traffic none > /dev/null 2> $OUT/w.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-replay --count-streams 1 --traffic none > /dev/null 2> $OUT/f.err
python3 - <<PY
import csv, glob, collections, re
for c, d in (('WRITE_SIZE', '$OUT/w'), ('FETCH_SIZE', '$OUT/f')):
    fs = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not fs:
        print(c, 'no csv'); continue
    acc = collections.defaultdict(float); n = collections.defaultdict(set)
    for row in csv.DictReader(open(fs[0])):
        m = re.search(r'(k_[a-z0-9_]+)', row['Kernel_Name'])
        if m and row['Counter_Name'] == c:
            acc[m.group(1)] += float(row['Counter_Value']); n[m.group(1)].add(row['Dispatch_Id'])
    print(c, {k: round(acc[k] * 1024 / len(n[k]) / 1e9, 3) for k in acc if acc[k] * 1024 / len(n[k]) > 5e7})
PY
cd $REPO
timeout 900 python3 bench.py --workload cfg4-band > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err; tail -c 300 $OUT/bench_cfg4.err
python3 - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_cfg4.json').read().strip().splitlines()[-1])
    print('cfg4-band ms/step', d['ms_per_step'], 'reads/s', d['value'], d['selfcheck'], d['downstream'])
    print({k: v for k, v in d['roofline']['kernels_ms_per_step'].items() if v > 5})
except Exception as e:
    print('cfg4 failed', e)
PY
