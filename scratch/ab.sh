#!/bin/bash
# A/B of library builds on ONE box: scratch/ab.sh <out-subdir> <bench args...> -- name=lib[:ENV=val,...] ...
# e.g. scratch/ab.sh ab1 --count-streams 1 -- r2=scratch/ab/libkv_r2.so new=kevlar_amd/libkvsketch_hip.so new_plain=kevlar_amd/libkvsketch_hip.so:KV_SKM_S2=plain
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$1; shift
mkdir -p "$OUT"
ARGS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARGS+=("$1"); shift; done
shift
for spec in "$@"; do
    name=${spec%%=*}; rest=${spec#*=}
    lib=${rest%%:*}; envs=""
    [ "$rest" != "$lib" ] && envs=${rest#*:}
    (
        export KV_LIB_PATH=$REPO/$lib
        [ -n "$envs" ] && export KV_TUNING=1          # (tuning switches are honoured only with it: kevlar_amd/csrc/kv_knobs.h)
        IFS=',' read -ra kv <<< "$envs"
        for e in "${kv[@]}"; do [ -n "$e" ] && export "$e"; done
        timeout 600 python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-replay --traffic none "${ARGS[@]}" > $OUT/$name.json 2> $OUT/$name.err
    )
    python3 - "$OUT/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d['roofline']['kernels_ms_per_step']
    keep = {n: round(v, 2) for n, v in k.items() if v >= 0.3}
    print(sys.argv[2], 'ms/step', d['ms_per_step'], 'hits', d['selfcheck']['hits_checksum'], keep, flush=True)
except Exception as exc:
    print(sys.argv[2], 'FAILED', exc, flush=True)
PY
done
