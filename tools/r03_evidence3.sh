#!/bin/bash
# Evidence for the final sources (GPU box): full -m gpu suite, headline profile set, turn scan counters, bench lines.
tag=${1:-r03_b}
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out/ev_$tag
mkdir -p $o
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $o/${tag}_pytest_gpu.log 2>&1; echo "pytest exit $?"
tail -3 $o/${tag}_pytest_gpu.log
bash tools/profile_round.sh $tag > $o/profile_round.log 2>&1; tail -3 $o/profile_round.log
cp $R/gpurun_out/prof_$tag/${tag}_hbm_traffic.json $R/profiles/hbm_traffic.json
bash tools/pmc_turn.sh $tag > $o/pmc_turn.log 2>&1
cd $R
timeout -k 10 100 python tools/turn_timing.py 250 2500 2 1000 3 lse valu > $o/${tag}_turn_timing_valu_form.log 2>&1
timeout -k 10 300 python bench.py > $o/${tag}_bench_f2.json.log 2> $o/bench_f2.err; echo "f2 exit $?"
timeout -k 10 300 python bench.py --workload ail --cpu-seconds 0 --no-iteration-probe > $o/${tag}_bench_ail.json.log 2> $o/bench_ail.err; echo "ail exit $?"
timeout -k 10 300 python bench.py --workload outbred --cpu-seconds 0 --no-iteration-probe > $o/${tag}_bench_outbred.json.log 2> $o/bench_outbred.err; echo "outbred exit $?"
timeout -k 10 300 python tools/iter_timing.py 2500 2500 4 2 0.013 flow det > $o/${tag}_iter_timing_config5_deterministic.log 2>&1; echo "det exit $?"
grep "^iteration\|^reserve" $o/${tag}_iter_timing_config5_deterministic.log
timeout -k 10 300 python tools/iter_timing.py 2500 2500 4 2 0.013 > $o/${tag}_iter_timing_config5.log 2>&1; echo "iter exit $?"
grep "^iteration\|^reserve\|^turn\|^sweep" $o/${tag}_iter_timing_config5.log
for f in $o/*bench*.json.log; do tail -1 $f | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print(j['value'], r['frac'], r['kernel_ms'], r['effective_clock_mhz'], r['valu_issue_frac'], r['frac_physical'])"; done
cat $R/gpurun_out/turn_$tag/kernel_stats.csv | cut -c1-150 | head -4
