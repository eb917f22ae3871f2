#!/bin/bash
# SQ counters of the super-k-mer kernels at config 2 (two passes of 8 counters), per wave-instruction totals
REPO=$(cd "$(dirname "$0")/.." && pwd)
# PMC_K=51: config 5's two-word keys; PMC_SCRIPT=scratch/pmc_band.py: the kernels of config 4's band shape; PMC_TAG names the output
SCRIPT=${PMC_SCRIPT:-scratch/pmc_count.py}
OUT=$REPO/gpurun_out/pmc_skm${PMC_TAG:+_$PMC_TAG}; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/a -- python3 $REPO/$SCRIPT > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/b -- python3 $REPO/$SCRIPT > $OUT/b.log 2>&1
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for sub in ('a', 'b'):
    files = glob.glob('$OUT/%s/**/*counter_collection.csv' % sub, recursive=True)
    seen = set()
    for row in csv.DictReader(open(files[0])):
        m = re.search(r'(k_[a-z0-9_]+)', row['Kernel_Name'])
        if not m: continue
        acc[m.group(1)][row['Counter_Name']] += float(row['Counter_Value'])
        if sub == 'a' and row['Counter_Name'] == 'SQ_WAVE_CYCLES': n[m.group(1)] += 1
for kname, d in sorted(acc.items()):
    if not d.get('SQ_WAVE_CYCLES'): continue
    wc = d['SQ_WAVE_CYCLES']
    print('%-22s launches %d  VALU %.3g  SALU %.3g  LDS %.3g  VMEM_RD %.3g  VMEM_WR %.3g  (wave-instructions, all launches)' % (
        kname, n[kname], d['SQ_INSTS_VALU'], d['SQ_INSTS_SALU'], d['SQ_INSTS_LDS'], d['SQ_INSTS_VMEM_RD'], d['SQ_INSTS_VMEM_WR']))
    print('%-22s of wave cycles: issuing %.0f%% (VALU %.0f%%, LDS %.0f%%, VMEM %.0f%%, scalar %.0f%%)  stalled on a busy pipe %.0f%% (LDS %.0f%%)  parked %.0f%%   LDS bank-conflict cycles %.3g' % (
        '', 100 * d['SQ_ACTIVE_INST_ANY'] / wc, 100 * d['SQ_ACTIVE_INST_VALU'] / wc, 100 * d['SQ_ACTIVE_INST_LDS'] / wc, 100 * d['SQ_ACTIVE_INST_VMEM'] / wc,
        100 * d['SQ_ACTIVE_INST_SCA'] / wc, 100 * d['SQ_WAIT_INST_ANY'] / wc, 100 * d['SQ_WAIT_INST_LDS'] / wc, 100 * d['SQ_WAIT_ANY'] / wc, d['SQ_LDS_BANK_CONFLICT']))
PY
