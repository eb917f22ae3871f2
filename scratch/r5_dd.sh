# record dedupe A/B: tests, then bench with and without on one box (alternating)
mkdir -p gpurun_out/r5_i
python -m pytest tests/test_gpu_skm.py tests/test_gpu_multicase.py -q -m gpu -x > gpurun_out/r5_i/t1.log 2>&1; tail -3 gpurun_out/r5_i/t1.log
for rep in 1 2; do
for dd in 1 0; do
KV_SKM_DEDUP=$dd KV_SKM_VERBOSE=1 python bench.py --steps 6 --warmup 2 > gpurun_out/r5_i/bench_dd$dd.json 2> gpurun_out/r5_i/bench_dd$dd.err
grep "kv_skm\] batch" gpurun_out/r5_i/bench_dd$dd.err | tail -2 | cut -c1-260
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r5_i/bench_dd$dd.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("dd=$dd", d["ms_per_step"], d["selfcheck"].get("hits_checksum"), {k:round(v,2) for k,v in r["kernels_ms_per_step"].items() if v>0.3})
except Exception as e:
    print("dd=$dd failed", e); print(open("gpurun_out/r5_i/bench_dd$dd.err").read()[-600:])
PY
done
done
