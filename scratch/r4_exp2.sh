#!/bin/bash
# round 4, experiment set 2 (one box): do the samples' kernels overlap better when the persistent kernels leave room on the CUs?
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd $REPO
L=kevlar_amd/libkvsketch_hip.so
bash scratch/ab.sh r4_exp2/cfg2_3s -- base=$L w3_2=$L:KV_SKM_WG3_PER_CU=2 w3_2_w1_2=$L:KV_SKM_WG3_PER_CU=2,KV_SKM_NWG1=512 w3_2_s15=$L:KV_SKM_WG3_PER_CU=2,KV_BIN_SLICE15=1 w3_2_w1_2_s15=$L:KV_SKM_WG3_PER_CU=2,KV_SKM_NWG1=512,KV_BIN_SLICE15=1 s15=$L:KV_BIN_SLICE15=1 base2=$L
bash scratch/ab.sh r4_exp2/cfg2_1s --count-streams 1 -- base=$L w3_2=$L:KV_SKM_WG3_PER_CU=2 s15=$L:KV_BIN_SLICE15=1
bash scratch/ab.sh r4_exp2/cfg2_2s --count-streams 2 -- base=$L w3_2=$L:KV_SKM_WG3_PER_CU=2
# what one rank of an 8-GPU run spends where (per-kernel times of the replay)
RANK_COST_PROF=1 RANK_COST_MODES=minimizer timeout 900 python3 scratch/exchange_rank_cost.py 8 2>&1 | tail -60
