#!/bin/bash
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/pmc_c; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/a -- python3 $REPO/scratch/pmc_count.py > $OUT/a.log 2>&1
python3 - <<PY
import csv, glob, collections, re
files = glob.glob('$OUT/a/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for row in csv.DictReader(open(files[0])):
    m = re.search(r'(k_[a-z0-9_]+)', row['Kernel_Name'])
    if m: acc[m.group(1)][row['Counter_Name']] += float(row['Counter_Value'])
for kname in ('k_bin_hash_direct', 'k_bin_split', 'k_bin_apply', 'k_novel_mark'):
    d = acc[kname]; items = 525e6 * (4 if kname != 'k_novel_mark' and kname != 'k_bin_hash_direct' else 1) * (3 if kname != 'k_novel_mark' else 1) / 64
    print(kname, 'VALU/wave-item %.0f SALU %.0f LDS %.1f  active %.0f%% wait_any %.0f%% wait_inst %.0f%%' % (d['SQ_INSTS_VALU'] / items, d['SQ_INSTS_SALU'] / items, d['SQ_INSTS_LDS'] / items, 100 * d['SQ_ACTIVE_INST_ANY'] / d['SQ_WAVE_CYCLES'], 100 * d['SQ_WAIT_ANY'] / d['SQ_WAVE_CYCLES'], 100 * d['SQ_WAIT_INST_ANY'] / d['SQ_WAVE_CYCLES']))
PY
