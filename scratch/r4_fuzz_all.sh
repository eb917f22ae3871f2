#!/bin/bash
# end-of-round randomised parity with fresh seeds (round 4: seeds 4xx)
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd $REPO
python3 -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
for spec in "fuzz_parity 160 401" "fuzz_list 160 402" "fuzz_shard 40 403" "fuzz_ingest 60 404" "fuzz_host 50 405" "fuzz_gunzip 80 406" "fuzz_kmer2bit 300 407"; do
  set -- $spec
  echo "== $1 ($2 trials, seed $3)"
  timeout 1500 python3 scratch/$1.py $2 $3 2>&1 | tail -3
  echo "rc $?"
done
