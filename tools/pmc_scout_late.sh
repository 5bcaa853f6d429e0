#!/bin/bash
# SQ counters of the scout kernels in the late regime of a run (tuning aid, GPU box): bash tools/pmc_scout_late.sh <tag> <lib|base>
tag=$1; v=${2:-base}
R=$GRAFT_REPO_ROOT
if [ "$v" = base ]; then lib=$R/cnf2freq_amd/libcnf2hip.so; else lib=$R/cnf2freq_amd/libcnf2hip_x_$v.so; fi
d=/tmp/pmc_$tag; rm -rf $d; mkdir -p $d; cp $lib $d/libcnf2hip.so; cp $R/cnf2freq_amd/libcnf2host.so $d/
export CNF2HIP_LIB=$d/libcnf2hip.so CNF2HOST_LIB=$d/libcnf2host.so CNF2_NO_STATS=1
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d/sq$i -- python3 $R/tools/flow_stats.py 100 1000 2 70 > $d/sq$i.log 2>&1 || echo "pass $i failed"
done
for k in haploweight_scout certainty_scout2 certainty_scout_kernel; do echo "== $k ($v)"; python3 $R/profiles/pmc_summarize.py $d/sq1 $d/sq2 --kernel "$k"; done > $R/gpurun_out/pmc_scout_late_$tag.txt
f=$(find $d/sq1 -name "*kernel_trace.csv" | head -1); python3 $R/tools/late_kernels.py $f 1.0 | head -6 >> $R/gpurun_out/pmc_scout_late_$tag.txt
cat $R/gpurun_out/pmc_scout_late_$tag.txt
