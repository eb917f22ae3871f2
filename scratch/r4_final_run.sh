#!/bin/bash
# the round's closing GPU call: build, the whole GPU suite, smoke(), then the evidence (profiles/collect.sh)
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r4_final_run; mkdir -p $OUT
cd $REPO
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 1500 python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
bash profiles/collect.sh r4_final > $OUT/collect.log 2>&1
tail -5 $OUT/pytest.log; tail -2 $OUT/smoke.log
