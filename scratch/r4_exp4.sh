#!/bin/bash
# S2's two speeds: does the stride between the buckets' write frontiers (segment capacity) move k_skm_split_sorted?
# (needs the two lines `if (getenv("KV_SKM_CAP1_EXTRA")) g.cap1 += ...; if (getenv("KV_SKM_CAP2_EXTRA")) g.cap2 += ...;` in skm_build, which were taken out
# again after the sweep -- no padding moved the kernel: 2.77-2.80 ms per step in all ten runs -- so that the kernel sources stay those of profiles/r4_final)
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd $REPO
L=kevlar_amd/libkvsketch_hip.so
bash scratch/ab.sh r4_exp4/cap2 --count-streams 1 -- base=$L c2_16=$L:KV_SKM_CAP2_EXTRA=16 c2_32=$L:KV_SKM_CAP2_EXTRA=32 c2_48=$L:KV_SKM_CAP2_EXTRA=48 c2_8=$L:KV_SKM_CAP2_EXTRA=8 c2_4=$L:KV_SKM_CAP2_EXTRA=4 c2_21=$L:KV_SKM_CAP2_EXTRA=21 c1_16=$L:KV_SKM_CAP1_EXTRA=16 c1_21=$L:KV_SKM_CAP1_EXTRA=21 base2=$L
