#!/bin/bash
# Evidence of a round for the final sources (GPU box):  bash tools/evidence.sh r04_f [parts]
#   parts (default "tests profile bench iter"):
#   tests    the -m gpu suite and smoke()                                   -> <tag>_pytest_gpu.log, trajectory_compared.txt
#   profile  headline profile set (tools/profile_round.sh: kernel stats, HBM counters, SQ counters)
#   bench    bench lines of every workload (the default line with its iteration probe; ail; outbred), the iteration bench on one
#            rank and as a 2-rank gloo rehearsal on this one GPU, bench.py --gpus 2 without a launcher
#   iter     rocprofv3 kernel stats of the iteration probe; CNF2_TIMING laps of config 5's setup and first iterations
#   cli      the executable on PlantImpute files of 5 000 individuals x 10 004 markers with CNF2_TIMING laps (tools/cli_scale.py)
#   config5  config 5 end to end, 100 iterations (tools/run_config5.py): ~3 minutes
#   c4       one of config 4's 8 shards on this GPU: 12 500 individuals x 200 080 markers (80 chromosomes x 2 501), 2 steps
# Everything lands in gpurun_out/ev_<tag>/; copy what is to be judged into profiles/.
tag=${1:-rXX}
parts=${2:-"tests profile bench iter"}
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out/ev_$tag
mkdir -p $o
cd $R
summ() { tail -1 $1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); r=j.get('roofline',{}); print(sys.argv[1], j['n_gpus'], j['value'], r.get('frac'), r.get('kernel_ms'), r.get('effective_clock_mhz'), r.get('valu_issue_frac'), r.get('frac_physical'), (j.get('cpu_baseline') or {}).get('value'), json.dumps(j.get('iteration_probe'))[:400])" $(basename $1); }
for part in $parts; do case $part in
tests)
    rm -f $R/gpurun_out/trajectory_compared.txt
    timeout -k 10 1100 python -m pytest tests -m gpu -q > $o/${tag}_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $o/${tag}_pytest_gpu.log
    cp $R/gpurun_out/trajectory_compared.txt $o/${tag}_trajectory_compared.txt
    timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.log 2>&1; echo "smoke exit $?"; tail -1 $o/smoke.log ;;
profile)
    bash tools/profile_round.sh $tag > $o/profile_round.log 2>&1; tail -3 $o/profile_round.log
    cp $R/gpurun_out/prof_$tag/${tag}_hbm_traffic.json $R/profiles/hbm_traffic.json ;;
bench)
    cd $R
    timeout -k 10 400 python bench.py > $o/${tag}_bench_f2.json.log 2> $o/bench_f2.err; echo "f2 exit $?"; summ $o/${tag}_bench_f2.json.log
    timeout -k 10 300 python bench.py --workload ail --cpu-seconds 0 --no-iteration-probe > $o/${tag}_bench_ail.json.log 2> $o/bench_ail.err; echo "ail exit $?"; summ $o/${tag}_bench_ail.json.log
    timeout -k 10 300 python bench.py --workload outbred --cpu-seconds 0 --no-iteration-probe > $o/${tag}_bench_outbred.json.log 2> $o/bench_outbred.err; echo "outbred exit $?"; summ $o/${tag}_bench_outbred.json.log
    timeout -k 10 300 python bench.py --gpus 2 --backend gloo --single-device --inds 2000 --no-iteration-probe > $o/${tag}_bench_rehearsal_2rank_gloo_single_device.json.log 2> $o/bench_2rank.err; echo "2-rank exit $?"; summ $o/${tag}_bench_rehearsal_2rank_gloo_single_device.json.log
    timeout -k 10 400 python bench.py --workload outbred --iterations 5 --warmup 2 > $o/${tag}_bench_iterations_outbred.json.log 2> $o/bench_it.err; echo "iterations exit $?"; summ $o/${tag}_bench_iterations_outbred.json.log
    timeout -k 10 400 python bench.py --gpus 2 --backend gloo --single-device --workload outbred --iterations 3 --warmup 1 --inds 2000 > $o/${tag}_bench_iterations_rehearsal_2rank_gloo_single_device.json.log 2> $o/bench_it2.err; echo "2-rank iterations exit $?"; summ $o/${tag}_bench_iterations_rehearsal_2rank_gloo_single_device.json.log ;;
iter)
    (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_probe -o probe -- python3 $R/tools/probe_iterations.py 500 2500 4 2 5 > $o/${tag}_iteration_probe.log 2>&1)
    python3 - $o $tag <<'PY'
import glob, sys
o, tag = sys.argv[1], sys.argv[2]
for f in glob.glob(o + "/prof_probe/**/*kernel_stats.csv", recursive=True):
    open("%s/%s_kernel_stats_iteration_probe_500fam_7it.csv" % (o, tag), "w").write("\n".join(open(f).read().split("\n")[:16]) + "\n")
PY
    cut -c1-160 $o/${tag}_kernel_stats_iteration_probe_500fam_7it.csv | head -12
    cd $R
    CNF2_TIMING=1 timeout -k 10 400 python tools/run_config5.py 2500 2500 4 6 > $o/${tag}_config5_setup_and_6_iterations_timing.log 2>&1; echo "timing exit $?"
    grep "postmarkerdata\|upload\|^iteration" $o/${tag}_config5_setup_and_6_iterations_timing.log | head -20 ;;
cli)
    cd $R
    timeout -k 10 500 python tools/cli_scale.py 500 4 2500 4 4 > $o/${tag}_cli_5000_individuals_x_10004_markers.log 2>&1; echo "cli exit $?"
    cat $o/${tag}_cli_5000_individuals_x_10004_markers.log | cut -c1-200 ;;
c4)
    cd $R
    timeout -k 10 600 python bench.py --inds 12500 --chroms 80 --steps 2 --warmup 1 --cpu-seconds 0 --no-iteration-probe --no-merge-probe > $o/${tag}_bench_c4_shard_12500x200080.json.log 2> $o/bench_c4.err; echo "c4 exit $?"; summ $o/${tag}_bench_c4_shard_12500x200080.json.log ;;
config5)
    cd $R
    timeout -k 10 700 python tools/run_config5.py 2500 2500 4 100 600 > $o/${tag}_config5_100_iterations.log 2>&1; echo "config5 exit $?"
    grep -v "^Scale factor\|^Number of" $o/${tag}_config5_100_iterations.log | tail -8 | cut -c1-600 ;;
esac; done
