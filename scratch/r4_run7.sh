#!/bin/bash
# the 2-bit per-k-mer kernels: parity, then cfg4-band (count + scan + downstream) and the default bench lines
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r4_run7; mkdir -p $OUT
cd $REPO
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 1500 python3 -m pytest tests/test_gpu_kmer2bit.py tests/test_gpu_skm.py tests/test_gpu_binned.py tests/test_gpu_cfg4_shape.py tests/test_gpu_sketch.py tests/test_gpu_pipeline.py tests/test_gpu_shard.py tests/test_golden_synth.py -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -25 $OUT/pytest.log
timeout 900 python3 bench.py --workload cfg4-band --no-downstream > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
python3 - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_cfg4.json').read().strip().splitlines()[-1])
    print('cfg4-band ms/step', d['ms_per_step'], 'reads/s', d['value'], d['selfcheck'])
    print({k: v for k, v in d['roofline']['kernels_ms_per_step'].items() if v > 5}, d['roofline']['host_wall_ms_per_step'])
except Exception as e:
    print('cfg4 failed', e)
PY
L=kevlar_amd/libkvsketch_hip.so
bash scratch/ab.sh r4_run7/cfg4_old --workload cfg4-band --no-downstream --steps 1 --warmup 1 -- old=$L:KV_BIN_2BIT=0,KV_NOVEL_2BIT=0
bash scratch/ab.sh r4_run7/cfg2 -- new=$L
