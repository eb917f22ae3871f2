#!/bin/bash
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r4_run8; mkdir -p $OUT
cd $REPO
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 1500 python3 -m pytest tests/test_gpu_kmer2bit.py tests/test_gpu_skm.py -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -30 $OUT/pytest.log
