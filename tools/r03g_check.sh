#!/bin/bash
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out/r03g
mkdir -p $o
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $o/pytest_gpu.log 2>&1; echo "pytest exit $?"
tail -4 $o/pytest_gpu.log
bash tools/pmc_turn.sh r03g
bash tools/profile_iter.sh r03g 0.19
cat $R/gpurun_out/prof_r03g/r03g_kernel_stats_config5_iteration.csv | cut -c1-160
