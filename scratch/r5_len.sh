# reads of 150 bases: the lane-per-read cut at two workgroups per CU against the wave kernel (KV_SKM_LANE_MAXWG=3: as before)
mkdir -p gpurun_out/r5_l
python -m pytest tests/test_gpu_skm.py -q -m gpu -x -k "other_lengths or count_matches" 2>&1 | tail -3
for rep in 1 2; do
for mw in 2 3; do
KV_SKM_LANE_MAXWG=$mw KV_SKM_VERBOSE=1 python bench.py --read-len ${LEN:-150} --steps 6 --warmup 2 --no-e2e --no-replay --traffic none --no-cpu-baseline > gpurun_out/r5_l/b.json 2> gpurun_out/r5_l/b.err
grep "batch of" gpurun_out/r5_l/b.err | tail -1 | cut -c1-200
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r5_l/b.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("maxwg=$mw", d["ms_per_step"], d["value"], d["selfcheck"].get("hits_checksum"), {k:round(v,2) for k,v in r["kernels_ms_per_step"].items() if v>0.3})
except Exception as e:
    print("maxwg=$mw failed", e); print(open("gpurun_out/r5_l/b.err").read()[-800:])
PY
done
done
