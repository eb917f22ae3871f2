#!/bin/bash
# ring size of the gzip decoder on 2 M reads (uniform and binned qualities)
for bits in ${BITS:-10 11 12}; do
  echo "== KV_GUNZIP_RING_BITS=$bits"
  KV_GUNZIP_RING_BITS=$bits timeout 300 python scratch/gunzip_rate.py 2000000 6 2>&1 | grep -E "qualities|k_gz|device" | awk 'NR==1||NR==6||NR==7||NR==8||NR==13||NR==14'
done
