mkdir -p gpurun_out/r5_i
for rep in 1 2; do
for mode in "1 1024" "0 0" "1 512"; do
set -- $mode
KV_SKM_DEDUP=$1 KV_SKM_DEDUP_RS=$2 KV_SKM_VERBOSE=1 python bench.py --steps 6 --warmup 2 --count-streams ${STREAMS:-3} --no-e2e --no-replay --traffic none --no-cpu-baseline > gpurun_out/r5_i/b.json 2> gpurun_out/r5_i/b.err
grep "batch of 525000000" gpurun_out/r5_i/b.err | tail -1 | cut -c60-260
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r5_i/b.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("mode=$mode", d["ms_per_step"], d["selfcheck"].get("hits_checksum"), {k:round(v,2) for k,v in r["kernels_ms_per_step"].items() if v>0.3})
except Exception as e:
    print("mode=$mode failed", e); print(open("gpurun_out/r5_i/b.err").read()[-600:])
PY
done
done
