#!/bin/bash
# per rank of config 2 (rank 0 replayed): the packing with the parallel scan, and fewer / wider S1 writers for the shards
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r4_exp8; mkdir -p $OUT
cd $REPO
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 900 python3 -m pytest tests/test_gpu_shard.py -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
for v in "768:512" "384:512" "256:512" "256:1024" "128:1024"; do
  KV_MEX_NWG1=${v%%:*} KV_SKM_S1_THREADS=${v#*:} RANK_COST_MODES=minimizer RANK_COST_PROF=1 timeout 600 python3 scratch/exchange_rank_cost.py 2 8 > $OUT/rank_cost_${v%%:*}_${v#*:}.log 2>&1
  echo "== writers ${v%%:*} threads ${v#*:}"; grep -h "^N=" $OUT/rank_cost_${v%%:*}_${v#*:}.log
  grep -h "k_skm_emit\|k_mex_pack\|k_skm_split\|k_skm_route " $OUT/rank_cost_${v%%:*}_${v#*:}.log | tail -4
done
