#!/bin/bash
# kernel durations of one-stream steps (rocprofv3 --kernel-trace --stats), averages by kernel: scratch/r6_trace.sh <out-subdir> [bench args]
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-replay --count-streams 1 --traffic none "$@" > $OUT/bench.json 2> $OUT/trace.err
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    print("%-72s calls %4s avg %9.1f us  %6s%%" % (row["Name"][:72], row["Calls"], float(row["AverageNs"]) / 1e3, row["Percentage"]))
PY
