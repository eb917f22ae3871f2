#!/bin/bash
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd $REPO
python3 -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
bash scratch/ab.sh r4_exp14/cfg2 --count-streams 1 -- base=kevlar_amd/libkvsketch_hip.so s15_cas=kevlar_amd/libkvsketch_hip.so:KV_BIN_SLICE15=1,KV_BIN_APPLY16=0 s15_sums=kevlar_amd/libkvsketch_hip.so:KV_BIN_SLICE15=1
KV_BIN_SLICE15=1 timeout 600 python3 -m pytest tests/test_gpu_skm.py -m gpu -q -k "sums_first or count_matches_oracle" 2>&1 | tail -3
