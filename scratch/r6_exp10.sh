#!/bin/bash
# round 6, experiment 10: k-mer occurrences per bucket around the default 8192 (a wave drains its share of a bucket's distinct k-mers 64 at a
# time: ~194 of them are three full rounds and a fourth for the last two) -- one stream, config 2
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd $REPO
L=kevlar_amd/libkvsketch_hip.so
scratch/ab.sh r6_exp10 --count-streams 1 -- b8192=$L b7600=$L:KV_SKM_BUCKET_KMERS=7600 b7000=$L:KV_SKM_BUCKET_KMERS=7000 b6500=$L:KV_SKM_BUCKET_KMERS=6500 b6000=$L:KV_SKM_BUCKET_KMERS=6000 b9000=$L:KV_SKM_BUCKET_KMERS=9000 b8192b=$L 2>&1 | cut -c1-200
