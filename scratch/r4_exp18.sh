#!/bin/bash
# the scan answered by the owners of the minimizer buckets: shard tests (2 / 3 ranks, declines, both scans), the bench launcher, then rank 0 replayed both ways
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r4_exp18; mkdir -p $OUT
cd $REPO
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 1800 python3 -m pytest tests/test_gpu_shard.py tests/test_bench_launcher.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -30 $OUT/pytest.log
for how in shard owner; do
  RANK_COST_SCAN=$how RANK_COST_MODES=minimizer RANK_COST_PROF=1 timeout 900 python3 scratch/exchange_rank_cost.py 2 4 8 > $OUT/rank_cost_$how.log 2>&1
  echo "== scan by $how"; grep -h "^N=" $OUT/rank_cost_$how.log; tail -4 $OUT/rank_cost_$how.log | grep -v "^N="
done
grep -h "k_skm_set_hits\|k_novel_mark_2bit\|k_skm_route " $OUT/rank_cost_owner.log | tail -6
