#!/bin/bash
# SQ counter passes for the sweep kernel on a 2000-individual slice (tuning aid; run on the GPU box).
# usage: bash tools/pmc_sq.sh <tag> [kernel-substring] [extra bench flags...]  -> gpurun_out/pmc_<tag>_summary.txt
tag=${1:-x}
kern=${2:-fb_fast}
shift; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- \
        python3 $R/bench.py --inds 2000 --steps 1 --warmup 0 --cpu-seconds 0 --no-iteration-probe "$@" > $out/p$i.log 2>&1 || echo "pass $i failed"
done
python3 $R/profiles/pmc_summarize.py $out --units $((2000*50020)) --kernel $kern > $R/gpurun_out/pmc_${tag}_summary.txt
cat $R/gpurun_out/pmc_${tag}_summary.txt
