#!/bin/bash
# where filter's and partition's seconds go on config 4's band output (cProfile of the two CLI stages)
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r4_exp11; mkdir -p $OUT
cd $REPO
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
KV_E2E_PROFILE=1 timeout 1200 python3 bench.py --workload cfg4-band --steps 1 --warmup 0 > $OUT/bench.json 2> $OUT/bench.err
grep -A40 "downstream profile" $OUT/bench.err | head -120
