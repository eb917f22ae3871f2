#!/bin/bash
# per rank of config 2 at N = 2, 4, 8 with the round's exchange path (all three item layouts), + the shard / launcher tests
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r4_exp9; mkdir -p $OUT
cd $REPO
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 1200 python3 -m pytest tests/test_gpu_shard.py tests/test_bench_launcher.py tests/test_gpu_kmer2bit.py -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
RANK_COST_PROF=1 timeout 900 python3 scratch/exchange_rank_cost.py 2 4 8 > $OUT/rank_cost.log 2>&1
grep -h "^N=" $OUT/rank_cost.log
