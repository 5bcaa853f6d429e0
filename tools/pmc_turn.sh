#!/bin/bash
# Kernel stats and SQ counters of the batched turn scan (GPU box): bash tools/pmc_turn.sh <tag>
#   -> gpurun_out/turn_<tag>/{kernel_stats.csv, pmc_sq_turn_rows_kernel.txt}; 250 families x 2500 SNPs x 2 chromosomes,
#      1 000 individuals (5.002e6 units per call)
tag=${1:-x}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/turn_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/turn_timing.py 250 2500 2 1000 3 > $out/timing.log 2>&1 || echo "timing failed"
cat $out/timing.log
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- \
    python3 $R/tools/turn_timing.py 250 2500 2 1000 3 > $out/stats.log 2>&1 || echo "stats pass failed"
f=$(find $out/stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -8 $f > $out/kernel_stats.csv && cat $out/kernel_stats.csv
[ "$2" = "nopmc" ] && exit 0
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_WAVES"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/sq$i -- \
        python3 $R/tools/turn_timing.py 250 2500 2 1000 1 > $out/sq$i.log 2>&1 || echo "sq pass $i failed"
done
python3 $R/profiles/pmc_summarize.py $out/sq1 $out/sq2 $out/sq3 $out/sq4 --units $((1000*5002)) --kernel turn_rows > $out/pmc_sq_turn_rows_kernel.txt
cat $out/pmc_sq_turn_rows_kernel.txt
