#!/bin/bash
# round 4, experiment set 1 (one box): S1 with 512- vs 1024-thread workgroups (time + WRITE_SIZE), k = 51 instances of k_skm_count
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r4_exp1; mkdir -p $OUT
cd $REPO
L=kevlar_amd/libkvsketch_hip.so
bash scratch/ab.sh r4_exp1/cfg2_1s --count-streams 1 -- s1_512=$L:KV_SKM_S1_THREADS=512 s1_1024=$L:KV_SKM_S1_THREADS=1024 s1_512b=$L:KV_SKM_S1_THREADS=512 s1_1024b=$L:KV_SKM_S1_THREADS=1024
bash scratch/ab.sh r4_exp1/cfg2_3s -- s1_512=$L:KV_SKM_S1_THREADS=512 s1_1024=$L:KV_SKM_S1_THREADS=1024 s1_512b=$L:KV_SKM_S1_THREADS=512 s1_1024b=$L:KV_SKM_S1_THREADS=1024
bash scratch/ab.sh r4_exp1/cfg5_1s --workload cfg5 --count-streams 1 -- anyk=$L:KV_SKM_ANY_K=1 k51=$L k51_w5=scratch/ab/libkv_k2w5.so k51_w4=scratch/ab/libkv_k2w4.so k51_w4_1024=scratch/ab/libkv_k2w4.so:KV_SKM_S1_THREADS=1024
bash scratch/ab.sh r4_exp1/cfg5_3s --workload cfg5 -- anyk=$L:KV_SKM_ANY_K=1 k51=$L k51_w5=scratch/ab/libkv_k2w5.so k51_w4=scratch/ab/libkv_k2w4.so
cd /tmp && export TMPDIR=/tmp
for t in 512 1024; do
  KV_SKM_S1_THREADS=$t timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w_$t -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-replay --count-streams 1 --traffic none > /dev/null 2> $OUT/w_$t.err
  KV_SKM_S1_THREADS=$t timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f_$t -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-replay --count-streams 1 --traffic none > /dev/null 2> $OUT/f_$t.err
  python3 - <<PY
import csv, glob, collections, re
for c, d in (('WRITE_SIZE', '$OUT/w_$t'), ('FETCH_SIZE', '$OUT/f_$t')):
    fs = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not fs:
        print('$t', c, 'no csv'); continue
    acc = collections.defaultdict(float); n = collections.defaultdict(set)
    for row in csv.DictReader(open(fs[0])):
        m = re.search(r'(k_[a-z0-9_]+)', row['Kernel_Name'])
        if m and row['Counter_Name'] == c:
            acc[m.group(1)] += float(row['Counter_Value']); n[m.group(1)].add(row['Dispatch_Id'])
    print('S1 threads $t', c, {k: round(acc[k] * 1024 / len(n[k]) / 1e9, 3) for k in acc if acc[k] * 1024 / len(n[k]) > 5e7})
PY
done
