#!/bin/bash
# One GPU-box call: full -m gpu suite, default bench, turn-scan timing.
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out/${1:-r03f}
mkdir -p $o
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $o/pytest_gpu.log 2>&1; echo "pytest exit $?"
tail -4 $o/pytest_gpu.log
timeout -k 10 400 python bench.py > $o/bench_default.json 2> $o/bench_default.err; echo "bench exit $?"
tail -1 $o/bench_default.json | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline'])"
bash tools/pmc_turn.sh ${1:-r03f} nopmc
