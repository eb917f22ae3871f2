#!/bin/bash
# end-of-round randomised parity with fresh seeds (round 5: seeds 5xx); FUZZ_SCALE / FUZZ_SEED as in r4_fuzz_all.sh
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd $REPO
python3 -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
for spec in "fuzz_parity ${FUZZ_SCALE:-1}60 ${FUZZ_SEED:-5}11" "fuzz_list ${FUZZ_SCALE:-1}60 ${FUZZ_SEED:-5}12" "fuzz_shard ${FUZZ_SCALE:-}40 ${FUZZ_SEED:-5}13" "fuzz_mex ${FUZZ_SCALE:-}40 ${FUZZ_SEED:-5}18" "fuzz_ingest ${FUZZ_SCALE:-}60 ${FUZZ_SEED:-5}14" "fuzz_host ${FUZZ_SCALE:-}50 ${FUZZ_SEED:-5}15" "fuzz_gunzip ${FUZZ_SCALE:-}80 ${FUZZ_SEED:-5}16" "fuzz_kmer2bit ${FUZZ_SCALE:-3}00 ${FUZZ_SEED:-5}17"; do
  set -- $spec
  echo "== $1 ($2 trials, seed $3)"
  timeout 1500 python3 scratch/$1.py $2 $3 2>&1 | grep -v amdgpu.ids | grep -v "^ok " | tail -6
  echo "rc $?"
done
