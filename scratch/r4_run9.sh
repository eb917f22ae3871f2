#!/bin/bash
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r4_run9; mkdir -p $OUT
cd $REPO
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 1200 python3 -m pytest tests/test_gpu_shard.py -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -30 $OUT/pytest.log
timeout 900 python3 scratch/fuzz_shard.py 40 11 > $OUT/fuzz_shard.log 2>&1; tail -3 $OUT/fuzz_shard.log
