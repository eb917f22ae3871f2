#!/bin/bash
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r4_exp16; mkdir -p $OUT
cd $REPO
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 1800 python3 -m pytest tests/test_bench_launcher.py tests/test_gpu_shard.py -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -25 $OUT/pytest.log
bash scratch/benchN.sh 4 8 > $OUT/benchN.log 2>&1; tail -12 $OUT/benchN.log
