#!/bin/bash
# Evidence, part 2 (GPU box): bench lines of every workload, the iteration bench (1 rank and a 2-rank gloo rehearsal on
# one GPU), config 5 end to end.   bash tools/r03_evidence2.sh r03_a
tag=${1:-r03_a}
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out/ev_$tag
mkdir -p $o
cd $R
timeout -k 10 300 python bench.py > $o/${tag}_bench_f2.json.log 2> $o/bench_f2.err; echo "f2 exit $?"
timeout -k 10 300 python bench.py --workload ail --cpu-seconds 0 --no-iteration-probe > $o/${tag}_bench_ail.json.log 2> $o/bench_ail.err; echo "ail exit $?"
timeout -k 10 300 python bench.py --workload outbred --cpu-seconds 0 --no-iteration-probe > $o/${tag}_bench_outbred.json.log 2> $o/bench_outbred.err; echo "outbred exit $?"
timeout -k 10 400 python bench.py --workload outbred --iterations 5 --warmup 2 > $o/${tag}_bench_iterations_outbred.json.log 2> $o/bench_iter.err; echo "iterations exit $?"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --backend gloo --single-device --workload outbred --iterations 3 --warmup 1 --inds 2000 \
    > $o/${tag}_bench_iterations_rehearsal_2rank_gloo_single_device.json.log 2> $o/bench_iter2.err; echo "rehearsal exit $?"
for f in $o/*.json.log; do echo "== $f"; tail -1 $f | cut -c1-600; done
timeout -k 10 700 python tools/run_config5.py 2500 2500 4 100 600 > $o/${tag}_config5_100_iterations.log 2>&1; echo "config5 exit $?"
grep -v "^Scale factor\|^Number of" $o/${tag}_config5_100_iterations.log | tail -12
