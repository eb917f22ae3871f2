#!/bin/bash
# chunk size of the block search on 2 M reads
for kb in ${KBS:-16 8 4}; do
  echo "== KV_GUNZIP_CHUNK_KB=$kb"
  KV_GUNZIP_CHUNK_KB=$kb timeout 300 python scratch/gunzip_rate.py 2000000 6 2>&1 | grep -E "k_gz|device" | sed -n "3,4p;9,10p"
done
