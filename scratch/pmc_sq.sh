#!/bin/bash
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/pmc_sq; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/a -- python3 $REPO/scratch/pmc_count.py > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/b -- python3 $REPO/scratch/pmc_count.py > $OUT/b.log 2>&1
python3 - <<PY
import csv, glob, collections, re
for sub in ('a', 'b'):
    files = glob.glob('$OUT/%s/**/*counter_collection.csv' % sub, recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(files[0])):
        m = re.search(r'(k_[a-z0-9_]+)', row['Kernel_Name'])
        if not m: continue
        acc[m.group(1)][row['Counter_Name']] += float(row['Counter_Value'])
    for kname, d in acc.items():
        print(sub, kname, {c: '%.3g' % v for c, v in sorted(d.items())})
PY
