#!/bin/bash
# SQ counters of the update pass's kernels over a short run in the steady state (tuning aid, GPU box):
#   bash tools/pmc_flow.sh <tag> [families] [iterations]  -> gpurun_out/pmc_flow_<tag>.txt
tag=${1:-x}
fams=${2:-200}
iters=${3:-36}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_flow_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAVES"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/sq$i -- \
        python3 $R/tools/flow_stats.py $fams 1000 2 $iters > $out/sq$i.log 2>&1 || echo "sq pass $i failed"
done
for k in certainty_scout_kernel certainty_scout2 certainty_finish haploweight_scout_kernel haploweight_scout2 haploweight_finish "guided_first_kernel<0" "guided_first_kernel<1" todo_; do
    echo "== $k"
    python3 $R/profiles/pmc_summarize.py $out/sq1 $out/sq2 $out/sq3 --kernel "$k"
done > $R/gpurun_out/pmc_flow_$tag.txt
cat $R/gpurun_out/pmc_flow_$tag.txt
