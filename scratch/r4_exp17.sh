#!/bin/bash
# config 5 (k = 51): k-mers per bucket against the loose list (k_skm_loose_count 1.44 + k_bin_spill 0.66 ms per step at the default)
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd $REPO
python3 -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
bash scratch/ab.sh r4_exp17/cfg5 --workload cfg5 --count-streams 1 -- default=kevlar_amd/libkvsketch_hip.so b2048=kevlar_amd/libkvsketch_hip.so:KV_SKM_BUCKET_KMERS=2048 b2560=kevlar_amd/libkvsketch_hip.so:KV_SKM_BUCKET_KMERS=2560 b3584=kevlar_amd/libkvsketch_hip.so:KV_SKM_BUCKET_KMERS=3584 b4096=kevlar_amd/libkvsketch_hip.so:KV_SKM_BUCKET_KMERS=4096
