#!/bin/bash
# config 5 (k = 51): k-mers per bucket against the loose list (k_skm_loose_count 1.44 + k_bin_spill 0.66 ms per step at the default)
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd $REPO
python3 -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
bash scratch/ab.sh r4_exp17/cfg5 --workload cfg5 --count-streams 1 -- default=kevlar_amd/libkvsketch_hip.so b1024=kevlar_amd/libkvsketch_hip.so:KV_SKM_BUCKET_KMERS=1024 b1536=kevlar_amd/libkvsketch_hip.so:KV_SKM_BUCKET_KMERS=1536 b1792=kevlar_amd/libkvsketch_hip.so:KV_SKM_BUCKET_KMERS=1792 b2048=kevlar_amd/libkvsketch_hip.so:KV_SKM_BUCKET_KMERS=2048
