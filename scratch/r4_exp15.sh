#!/bin/bash
# S2 with the next chunk's records requested ahead: parity, then the step (k_skm_split was 2.78-2.79 ms per step of config 2, 2.35-2.37 of config 5, on every box)
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r4_exp15; mkdir -p $OUT
cd $REPO
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 1500 python3 -m pytest tests/test_gpu_skm.py tests/test_gpu_shard.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
for rep in 1 2; do bash scratch/ab.sh r4_exp15/cfg2_$rep --count-streams 1 -- new=kevlar_amd/libkvsketch_hip.so; done
bash scratch/ab.sh r4_exp15/cfg2_3s -- new=kevlar_amd/libkvsketch_hip.so
bash scratch/ab.sh r4_exp15/cfg5 --workload cfg5 --count-streams 1 -- new=kevlar_amd/libkvsketch_hip.so
