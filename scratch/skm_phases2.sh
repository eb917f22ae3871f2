#!/bin/bash
# phase dissection of k_skm_count for two library builds on one box
REPO=$(cd "$(dirname "$0")/.." && pwd)
for lib in "$@"; do
  echo "== $lib"
  KV_LIB_PATH=$REPO/$lib python3 $REPO/scratch/skm_phases.py 0 2 128 256 1 64 2>&1 | grep -v "^scan failed" | tail -8
done
