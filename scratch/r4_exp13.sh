#!/bin/bash
# k_bin_apply_w16 (16-bit sums, saturation once): parity tests, then A/B against the compare-and-swap kernel at config 2 and config 5
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r4_exp13; mkdir -p $OUT
cd $REPO
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 1500 python3 -m pytest tests/test_gpu_skm.py tests/test_gpu_binned.py tests/test_gpu_fullsize.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
for rep in 1 2; do
bash scratch/ab.sh r4_exp13/cfg2_$rep --count-streams 1 -- cas=kevlar_amd/libkvsketch_hip.so:KV_BIN_APPLY16=0 sums=kevlar_amd/libkvsketch_hip.so
done
bash scratch/ab.sh r4_exp13/cfg2_3s -- cas=kevlar_amd/libkvsketch_hip.so:KV_BIN_APPLY16=0 sums=kevlar_amd/libkvsketch_hip.so
bash scratch/ab.sh r4_exp13/cfg5 --workload cfg5 --count-streams 1 -- cas=kevlar_amd/libkvsketch_hip.so:KV_BIN_APPLY16=0 sums=kevlar_amd/libkvsketch_hip.so
