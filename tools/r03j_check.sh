#!/bin/bash
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out/r03j
mkdir -p $o
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -q -k "engine or dist" > $o/pytest.log 2>&1; echo "pytest exit $?"; tail -4 $o/pytest.log
timeout -k 10 300 python tools/ab_update_exact.py 200 1000 2 45 > $o/ab_update_exact_200fam.log 2>&1; echo "ab exit $?"; tail -2 $o/ab_update_exact_200fam.log
timeout -k 10 700 python tools/run_config5.py 2500 2500 4 100 600 > $o/config5_100_iterations.log 2>&1; echo "config5 exit $?"
grep -v "^Scale factor\|^Number of" $o/config5_100_iterations.log | tail -14 | cut -c1-700
