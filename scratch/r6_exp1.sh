#!/bin/bash
# round 6, experiment 1: (a) issue rates of the drain's integer instructions; (b) do differently-bound kernels of the three count streams
# overlap better when S3 leaves room on the CU (KV_SKM_WG3_PER_CU = 2 / 1)?
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r6_exp1; mkdir -p $OUT
cd $REPO
scratch/micro/valu_rates > $OUT/valu_rates.txt 2>&1; cat $OUT/valu_rates.txt
L=kevlar_amd/libkvsketch_hip.so
scratch/ab.sh r6_exp1 -- base=$L wg2=$L:KV_TUNING=1,KV_SKM_WG3_PER_CU=2 wg1=$L:KV_TUNING=1,KV_SKM_WG3_PER_CU=1 base2=$L
scratch/ab.sh r6_exp1/one --count-streams 1 -- base=$L wg2=$L:KV_TUNING=1,KV_SKM_WG3_PER_CU=2
