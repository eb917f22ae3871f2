#!/bin/bash
# SQ counters of the inflate kernels on one gzip stream of 2 M reads (two passes of 8 counters)
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/pmc_gunzip; rm -rf $OUT; mkdir -p $OUT
cd $REPO
python3 scratch/ingest_phases.py 2000000 gzip > $OUT/make.log 2>&1        # leaves /tmp/phases.fq.gz
export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/a -- python3 scratch/gunzip_file.py /tmp/phases.fq.gz > $OUT/a.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_INSTS_SMEM --output-format csv -d $OUT/b -- python3 scratch/gunzip_file.py /tmp/phases.fq.gz > $OUT/b.log 2>&1
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for sub in ('a', 'b'):
    files = glob.glob('$OUT/%s/**/*counter_collection.csv' % sub, recursive=True)
    for row in csv.DictReader(open(files[0])):
        m = re.search(r'(k_[a-z0-9_]+)', row['Kernel_Name'])
        if not m: continue
        acc[m.group(1)][row['Counter_Name']] += float(row['Counter_Value'])
        if sub == 'a' and row['Counter_Name'] == 'SQ_WAVE_CYCLES': n[m.group(1)] += 1
for kname, d in sorted(acc.items()):
    if not d.get('SQ_WAVE_CYCLES'): continue
    wc = d['SQ_WAVE_CYCLES']
    print('%-16s launches %d  wave-cycles %.3g  VALU %.3g  SALU %.3g  SMEM %.3g  LDS %.3g  VMEM_RD %.3g  VMEM_WR %.3g  (wave-instructions, all launches)' % (
        kname, n[kname], wc, d['SQ_INSTS_VALU'], d['SQ_INSTS_SALU'], d['SQ_INSTS_SMEM'], d['SQ_INSTS_LDS'], d['SQ_INSTS_VMEM_RD'], d['SQ_INSTS_VMEM_WR']))
    print('%-16s of wave cycles: issuing %.0f%% (VALU %.0f%%, LDS %.0f%%, VMEM %.0f%%, scalar %.0f%%)  stalled on a busy pipe %.0f%% (LDS %.0f%%)  parked %.0f%%' % (
        '', 100 * d['SQ_ACTIVE_INST_ANY'] / wc, 100 * d['SQ_ACTIVE_INST_VALU'] / wc, 100 * d['SQ_ACTIVE_INST_LDS'] / wc, 100 * d['SQ_ACTIVE_INST_VMEM'] / wc,
        100 * d['SQ_ACTIVE_INST_SCA'] / wc, 100 * d['SQ_WAIT_INST_ANY'] / wc, 100 * d['SQ_WAIT_INST_LDS'] / wc, 100 * d['SQ_WAIT_ANY'] / wc))
PY
