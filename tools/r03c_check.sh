#!/bin/bash
# One GPU-box call: full -m gpu suite, the exact A/B of the update kernels, the default bench and the iteration bench.
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out/r03c
mkdir -p $o
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $o/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $o/pytest_gpu.log
tail -5 $o/pytest_gpu.log
timeout -k 10 300 python tools/ab_update_exact.py 200 1000 2 45 > $o/ab_update_exact_200fam.log 2>&1; echo "ab exit $?"
tail -5 $o/ab_update_exact_200fam.log
timeout -k 10 400 python bench.py > $o/bench_default.json 2> $o/bench_default.err; echo "bench exit $?"
tail -2 $o/bench_default.json
timeout -k 10 300 python bench.py --workload outbred --iterations 3 --inds 2000 --warmup 1 > $o/bench_iter_2000.json 2> $o/bench_iter_2000.err; echo "bench iter exit $?"
tail -2 $o/bench_iter_2000.json
