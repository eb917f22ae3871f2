#!/bin/bash
# A/B on one box: ASCII expansion of the hashed k-mers through the LDS byte table (default build) against v_perm_b32 over the
# constant "ACGT" (-DKV_ASCII_PERM, scratch/ab/libkv_perm.so built beforehand): config 2 back to back, config 5, config 4's band step
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd $REPO
python3 -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
for rep in 1 2; do
bash scratch/ab.sh r4_exp6/cfg2_$rep --count-streams 1 -- table=kevlar_amd/libkvsketch_hip.so perm=scratch/ab/libkv_perm.so
done
bash scratch/ab.sh r4_exp6/cfg2_3streams -- table=kevlar_amd/libkvsketch_hip.so perm=scratch/ab/libkv_perm.so
bash scratch/ab.sh r4_exp6/cfg5 --workload cfg5 --count-streams 1 -- table=kevlar_amd/libkvsketch_hip.so perm=scratch/ab/libkv_perm.so
for v in table:kevlar_amd/libkvsketch_hip.so perm:scratch/ab/libkv_perm.so; do
  KV_LIB_PATH=$REPO/${v#*:} timeout 600 python3 bench.py --workload cfg4-band --steps 1 --warmup 1 --no-downstream > gpurun_out/r4_exp6/cfg4_${v%%:*}.json 2> gpurun_out/r4_exp6/cfg4_${v%%:*}.err
  python3 -c "
import json,sys; d=json.loads(open('gpurun_out/r4_exp6/cfg4_${v%%:*}.json').read().strip().splitlines()[-1]); k=d['roofline']['kernels_ms_per_step']; print('cfg4 ${v%%:*}', d['ms_per_step'], d['selfcheck']['hits_checksum'], {n: round(x,1) for n,x in k.items() if x>50})"
done
