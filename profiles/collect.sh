#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box:
#   gpurun --timeout 1500 -- 'bash profiles/collect.sh r1_final'
# Outputs under gpurun_out/<round>/ (scratch); profiles/summarise.py turns them into the small files
# committed under profiles/<round>/.  Three separate profiler passes: kernel trace + stats, then one
# --pmc pass per counter (FETCH_SIZE and WRITE_SIZE do not fit one pass; counters are never combined
# with trace domains other than the kernel trace).
set -u
ROUND=${1:-r4_final}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$ROUND
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# the default bench counts the three samples concurrently on three streams; the trace and the counter passes run them
# back to back (--count-streams 1) so that a kernel's duration and counters are its own
BENCH_ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-replay --count-streams 1 --traffic none"
timeout 900 python3 $REPO/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err          # the driver's command
timeout 600 python3 $REPO/bench.py --steps 20 --warmup 5 --count-streams 1 --no-cpu-baseline --no-e2e --no-replay --traffic none > $OUT/bench_one_stream.json 2> $OUT/bench_one_stream.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $BENCH_ARGS > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_default -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-replay --traffic none > $OUT/bench_under_rocprof_default.json 2> $OUT/trace_default.err
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-replay --count-streams 1 --traffic none > /dev/null 2> $OUT/pmc_fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-replay --count-streams 1 --traffic none > /dev/null 2> $OUT/pmc_write.err
# ingest: one ordinary gzip stream / one BGZF file of 2 M reads through the device inflaters   (COLLECT_SKIP="ingest cfg5" leaves parts out)
[[ " ${COLLECT_SKIP:-} " == *" ingest "* ]] || (cd $REPO && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_gunzip -- python3 scratch/gunzip_rate.py 2000000 6 > $OUT/gunzip_rate.log 2> $OUT/trace_gunzip.err)
[[ " ${COLLECT_SKIP:-} " == *" ingest "* ]] || (cd $REPO && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_bgzf -- python3 scratch/inflate_rate.py 2000000 > $OUT/inflate_rate.log 2> $OUT/trace_bgzf.err)
# SQ counters of the count / scan kernels at their current shape (two --pmc passes of eight counters over one count of
# each sample + one scan)
(cd $REPO && bash scratch/pmc_skm.sh > $OUT/pmc_skm.log 2>&1)
# the same for config 5 (k = 51: two-word keys) and for the kernels of config 4's band shape (k_consume under banding, k_novel_mark)
[[ " ${COLLECT_SKIP:-} " == *" cfg5 "* ]] || (cd $REPO && PMC_K=51 PMC_TAG=cfg5 bash scratch/pmc_skm.sh > $OUT/pmc_skm_cfg5.log 2>&1)
[[ " ${COLLECT_SKIP:-} " == *" cfg4 "* ]] || (cd $REPO && PMC_SCRIPT=scratch/pmc_band.py PMC_TAG=cfg4 bash scratch/pmc_skm.sh > $OUT/pmc_skm_cfg4.log 2>&1)
# config 5 (proband + 3 controls, k = 51): bench line and the profiler's kernel statistics
[[ " ${COLLECT_SKIP:-} " == *" cfg5 "* ]] || timeout 1200 python3 $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --workload cfg5 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err      # the driver's style of command
[[ " ${COLLECT_SKIP:-} " == *" cfg5 "* ]] || timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg5 -- python3 $REPO/bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-replay --count-streams 1 > $OUT/bench_cfg5_under_rocprof.json 2> $OUT/trace_cfg5.err
# config 4 as one of its eight GPUs sees it: the step with its downstream stages, and one step under the tracer
[[ " ${COLLECT_SKIP:-} " == *" cfg4 "* ]] || timeout 1200 python3 $REPO/bench.py --workload cfg4-band > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
[[ " ${COLLECT_SKIP:-} " == *" cfg4 "* ]] || timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg4 -- python3 $REPO/bench.py --workload cfg4-band --steps 1 --warmup 1 --no-downstream --count-streams 1 > $OUT/bench_cfg4_under_rocprof.json 2> $OUT/trace_cfg4.err
python3 $REPO/profiles/summarise.py $OUT $OUT/summary
cp $REPO/gpurun_out/pmc_skm/summary.txt $OUT/summary/sq_counters.txt 2>/dev/null
cp $REPO/gpurun_out/pmc_skm_cfg5/summary.txt $OUT/summary/sq_counters_cfg5.txt 2>/dev/null
cp $REPO/gpurun_out/pmc_skm_cfg4/summary.txt $OUT/summary/sq_counters_cfg4_band.txt 2>/dev/null
ls -la $OUT/summary
