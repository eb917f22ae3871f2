#!/bin/bash
# what a rank of the BANDED layout computes per step of config 2 at N = 2, 4, 8 (one GPU plays band 0; samples back to back as a rank counts them)
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r4_exp10; mkdir -p $OUT
cd $REPO
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
for n in 1 2 4 8; do
  extra=""; [ $n -gt 1 ] && extra="--bands $n"
  timeout 600 python3 bench.py --steps 10 --warmup 3 --count-streams 1 --no-cpu-baseline --no-e2e --no-replay --traffic none $extra > $OUT/band_$n.json 2> $OUT/band_$n.err
  python3 -c "
import json; d=json.loads(open('$OUT/band_$n.json').read().strip().splitlines()[-1]); k=d['roofline']['kernels_ms_per_step']
print('bands $n', d['ms_per_step'], d['roofline']['host_wall_ms_per_step'], {a: round(b,2) for a,b in k.items() if b>=0.3})"
done
