#!/bin/bash
# VALU / SALU / LDS wave-instructions of k_skm_count with parts of it switched off (KV_SKM_DEBUG): scratch/pmc_dbg.sh 0 1 64 ...
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/pmc_dbg; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for dbg in "$@"; do
    export KV_SKM_DEBUG=$dbg
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/d$dbg -- python3 $REPO/scratch/pmc_count.py > $OUT/d$dbg.log 2>&1
    python3 - $OUT/d$dbg $dbg <<'PY'
import csv, glob, collections, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
for row in csv.DictReader(open(f)):
    m = re.search(r'(k_[a-z0-9_]+)', row['Kernel_Name'])
    if m: acc[m.group(1)][row['Counter_Name']] += float(row['Counter_Value'])
for kname in ('k_skm_count', 'k_skm_novel', 'k_skm_emit_wave'):
    d = acc.get(kname)
    if d: print('dbg', sys.argv[2], kname, 'VALU %.3g SALU %.3g LDS %.3g  VALU active %.0f%% of wave cycles' % (d['SQ_INSTS_VALU'], d['SQ_INSTS_SALU'], d['SQ_INSTS_LDS'], 100 * d['SQ_ACTIVE_INST_VALU'] / max(1, d['SQ_WAVE_CYCLES'])), flush=True)
PY
done
