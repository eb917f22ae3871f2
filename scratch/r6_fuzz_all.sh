#!/bin/bash
# end-of-round randomised parity with fresh seeds (round 6: seeds 6xx; the fuzzers pin kernel paths through tuning switches, which need KV_TUNING=1)
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd $REPO
export KV_TUNING=1
python3 -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
for spec in "fuzz_parity ${FUZZ_SCALE:-1}60 ${FUZZ_SEED:-6}01" "fuzz_list ${FUZZ_SCALE:-1}60 ${FUZZ_SEED:-6}02" "fuzz_shard ${FUZZ_SCALE:-}40 ${FUZZ_SEED:-6}03" "fuzz_mex ${FUZZ_SCALE:-}60 ${FUZZ_SEED:-6}08" "fuzz_ingest ${FUZZ_SCALE:-}60 ${FUZZ_SEED:-6}04" "fuzz_host ${FUZZ_SCALE:-}50 ${FUZZ_SEED:-6}05" "fuzz_gunzip ${FUZZ_SCALE:-}80 ${FUZZ_SEED:-6}06" "fuzz_kmer2bit ${FUZZ_SCALE:-3}00 ${FUZZ_SEED:-6}07"; do
  set -- $spec
  echo "== $1 ($2 trials, seed $3)"
  timeout 1500 python3 scratch/$1.py $2 $3 2>&1 | tail -3
  echo "rc $?"
done
