#!/bin/bash
# Final check of the round (GPU box): the -m gpu suite, smoke(), the default bench line, config 5 end to end.
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out/ev_r03_f
mkdir -p $o
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $o/r03_f_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -2 $o/r03_f_pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $o/smoke.log
timeout -k 10 300 python bench.py > $o/r03_f_bench_f2.json.log 2> $o/bench.err; echo "bench exit $?"
tail -1 $o/r03_f_bench_f2.json.log | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print(j['value'], r['frac'], r['kernel_ms'], r['effective_clock_mhz'], r['valu_issue_frac'], r['frac_physical'], j['cpu_baseline']['value'])"
timeout -k 10 700 python tools/run_config5.py 2500 2500 4 100 600 > $o/r03_f_config5_100_iterations.log 2>&1; echo "config5 exit $?"
grep -v "^Scale factor\|^Number of" $o/r03_f_config5_100_iterations.log | tail -8 | cut -c1-500
