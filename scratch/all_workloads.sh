#!/bin/bash
# the default bench line of every workload (CPU baseline and end-to-end legs included)
for w in cfg1 cfg5 cfg4-proxy; do
  s=$(date +%s)
  timeout 900 python bench.py --workload $w 2>/tmp/err_$w.txt > /tmp/out_$w.json
  echo "$w rc=$? $(( $(date +%s) - s )) s"
  python3 - "$w" <<'PY'
import sys, json
d = json.loads(open('/tmp/out_%s.json' % sys.argv[1]).read().strip().split('\n')[-1])
print('  ', d['ms_per_step'], 'ms/step', round(d['value'] / 1e6, 1), 'M reads/s; cpu', (d.get('cpu_baseline') or {}).get('value'), '; e2e', (d.get('end_to_end') or {}).get('value'), '; frac', d['roofline']['frac'], d['selfcheck'].get('hits_checksum'))
PY
done
