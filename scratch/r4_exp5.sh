#!/bin/bash
# the list scan's first probe from a bit map of the case sample's table 0 (KV_NOVEL_BITS=1) against the table itself
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd $REPO
L=kevlar_amd/libkvsketch_hip.so
bash scratch/ab.sh r4_exp5/cfg2_1s --count-streams 1 -- base=$L bits=$L:KV_NOVEL_BITS=1 base2=$L bits2=$L:KV_NOVEL_BITS=1
bash scratch/ab.sh r4_exp5/cfg2_3s -- base=$L bits=$L:KV_NOVEL_BITS=1
bash scratch/ab.sh r4_exp5/cfg5_1s --workload cfg5 --count-streams 1 -- base=$L bits=$L:KV_NOVEL_BITS=1
