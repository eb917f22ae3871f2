#!/bin/bash
# premise test: how much of k_skm_count is probe depth?  Smaller buckets = emptier LDS tables = shorter probe chains (more buckets though)
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd $REPO
L=kevlar_amd/libkvsketch_hip.so
bash scratch/ab.sh r4_exp3/cfg2_1s --count-streams 1 -- base=$L b6144=$L:KV_SKM_BUCKET_KMERS=6144 b4096=$L:KV_SKM_BUCKET_KMERS=4096 b3072=$L:KV_SKM_BUCKET_KMERS=3072 b12288=$L:KV_SKM_BUCKET_KMERS=12288
