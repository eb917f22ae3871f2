#!/bin/bash
# round 5: the GPU's clock and power while the step runs: bench.py in the background, rocm-smi sampled beside it
python bench.py --steps 1500 --warmup 5 --traffic none --no-cpu-baseline --no-e2e --no-replay > /tmp/clock_bench.json 2>/dev/null &
pid=$!
sleep 30
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -i "sclk\|mclk\|power\|busy\|fclk" | tr -s ' ' | head -8
  echo ---
  sleep 1
done
wait $pid
python -c "
import json
d=json.loads(open('/tmp/clock_bench.json').read().strip().split('\n')[-1]); print('ms_per_step', d['ms_per_step'])"
/opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | tr -s ' ' | head -4
