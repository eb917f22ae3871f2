#!/bin/bash
# set scan through the 2-bit per-k-mer kernel against the bucketed one, per rank of config 2 (rank 0 replayed on this GPU), + the shard tests
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r4_exp7; mkdir -p $OUT
cd $REPO
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 900 python3 -m pytest tests/test_gpu_shard.py -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
for how in skm 2bit; do
  KV_SET_SCAN=$how RANK_COST_MODES=minimizer RANK_COST_PROF=1 timeout 900 python3 scratch/exchange_rank_cost.py 2 4 8 > $OUT/rank_cost_$how.log 2>&1
done
tail -5 $OUT/pytest.log
grep -h "total\|ms per step\|N =" $OUT/rank_cost_skm.log | head -20
grep -h "total\|ms per step\|N =" $OUT/rank_cost_2bit.log | head -20
