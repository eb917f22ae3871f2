#!/bin/bash
# WRITE_SIZE and duration of the two record-moving kernels under different writer counts
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/pmc_w; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "768 4" "512 4" "512 2" "256 2" "768 2" "768 8"; do
  set -- $cfg
  export KV_SKM_NWG1=$1 KV_SKM_NWG2=$2
  python3 $REPO/scratch/skm_phases.py 0 2>/dev/null | tail -1 | sed "s/^/nwg1=$1 nwg2=$2 time /"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w_$1_$2 -- python3 $REPO/scratch/pmc_count.py > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections, re
f = glob.glob('$OUT/w_$1_$2/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(float); n = collections.Counter()
for row in csv.DictReader(open(f)):
    m = re.search(r'(k_skm_emit|k_skm_split)', row['Kernel_Name'])
    if m and row['Counter_Name'] == 'WRITE_SIZE':
        acc[m.group(1)] += float(row['Counter_Value']); 
        n[(m.group(1), row['Dispatch_Id'])] += 1
for k in acc:
    launches = len([1 for (kk, d) in n if kk == k])
    print('   ', k, 'WRITE_SIZE GB per launch %.2f' % (acc[k] * 1024 / launches / 1e9))
PY
done
