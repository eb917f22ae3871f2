#!/bin/bash
# round 6, experiment 9: the per-k-mer kernels of config 4's band shape (k_bin_hash_2bit, k_novel_mark_2bit) compiled for k = 31 with the
# product-table murmurs, against the build before
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r6_exp9; mkdir -p $OUT
cd $REPO
timeout 900 python3 -m pytest tests/test_gpu_kmer2bit.py tests/test_gpu_binned.py tests/test_gpu_cfg4_shape.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
for v in pre2bit new pre2bit new; do
  lib=kevlar_amd/libkvsketch_hip.so; [ $v = pre2bit ] && lib=scratch/ab/libkv_pre2bit.so
  KV_LIB_PATH=$REPO/$lib timeout 900 python3 bench.py --workload cfg4-band --steps 2 --warmup 1 --no-downstream --no-cpu-baseline --no-e2e --no-replay --traffic none > $OUT/$v.json 2> $OUT/$v.err
  python3 - $OUT/$v.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d['roofline']['kernels_ms_per_step']
print(sys.argv[2], 'ms/step', d['ms_per_step'], 'hits', d['selfcheck']['hits_checksum'], {n: round(v, 1) for n, v in k.items() if v > 20})
PY
done
