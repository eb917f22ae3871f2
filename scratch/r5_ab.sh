#!/bin/bash
# round 5: A/B of environment variants of one library on ONE box: scratch/r5_ab.sh OUTDIR "VAR=val VAR2=val" ...  (kernels back to back:
# --count-streams 1, so the per-kernel HIP-event times are clean; then the driver's three-stream step)
out=$1; shift
mkdir -p $out
i=0
for spec in "$@"; do
  i=$((i+1))
  for streams in 1 3; do
    env $spec python bench.py --traffic none --no-cpu-baseline --no-e2e --no-replay --count-streams $streams > $out/b_${i}_s$streams.json 2> $out/b_${i}_s$streams.err
    python - "$spec" $streams $out/b_${i}_s$streams.json <<'PY'
import json,sys
spec,streams,f=sys.argv[1:4]
try:
    d=json.loads(open(f).read().strip().split("\n")[-1])
    ks={k:round(x,2) for k,x in d["roofline"]["kernels_ms_per_step"].items() if x>0.25}
    print("[%s] streams=%s %.3f ms/step %s %s" % (spec, streams, d["ms_per_step"], d["selfcheck"]["hits_checksum"], ks if streams=="1" else ""))
except Exception as e:
    print("[%s] streams=%s FAILED %r" % (spec, streams, e)); print(open(f.replace('.json','.err')).read()[-800:])
PY
  done
done
