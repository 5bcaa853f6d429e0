#!/bin/bash
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out/r03h
mkdir -p $o
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -q -k "turn or reserve" > $o/pytest_turn.log 2>&1; echo "pytest exit $?"
tail -15 $o/pytest_turn.log
bash tools/pmc_turn.sh r03h
timeout -k 10 100 python tools/turn_timing.py 250 2500 2 1000 3 lse valu
