#!/bin/bash
# SQ counters of the accumulate kernel (tuning aid, GPU box): bash tools/pmc_acc.sh <tag> [kernel substring]
# 250 families x 2500 SNPs x 2 chromosomes, one iteration -> gpurun_out/pmc_acc_<tag>.txt (per (individual, marker) unit)
tag=${1:-x}
kern=${2:-acc_paths}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_acc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_WAVES"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/sq$i -- \
        python3 $R/tools/iter_timing.py 250 2500 2 1 > $out/sq$i.log 2>&1 || echo "sq pass $i failed"
done
python3 $R/profiles/pmc_summarize.py $out/sq1 $out/sq2 $out/sq3 $out/sq4 --units $((1000*5002)) --kernel $kern > $R/gpurun_out/pmc_acc_$tag.txt
cat $R/gpurun_out/pmc_acc_$tag.txt
