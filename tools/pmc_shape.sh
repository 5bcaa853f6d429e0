#!/bin/bash
# SQ counters of the sweep kernel on the F2 and the outbred workload side by side (tuning aid, GPU box):
#   bash tools/pmc_shape.sh   -> per-kernel sums on stdout (2000 individuals each; divide by individuals x markers)
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_shape
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for wl in ${PMC_WORKLOADS:-f2 outbred}; do
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$wl$i -- \
        python3 $R/bench.py --workload $wl --inds 2000 --steps 1 --warmup 0 --cpu-seconds 0 --no-iteration-probe --no-merge-probe > $out/$wl$i.log 2>&1 || echo "pass failed"
done
done
for wl in ${PMC_WORKLOADS:-f2 outbred}; do
  echo "== $wl"
  python3 $R/profiles/pmc_summarize.py $out/${wl}1 $out/${wl}2 $out/${wl}3 --kernel fb_fast
done
