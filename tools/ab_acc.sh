#!/bin/bash
# A/B timing of library builds on the accumulate path (tuning aid, GPU box): bash tools/ab_acc.sh name1 name2 ...
# -> sweep+accumulate time of tools/iter_timing.py (2500 families x 2500 SNPs x 2 chromosomes) with
# cnf2freq_amd/libcnf2hip_x_<name>.so in place of the product library ("base" = the product library)
for v in "$@"; do
    if [ "$v" = base ]; then lib=$PWD/cnf2freq_amd/libcnf2hip.so; else lib=$PWD/cnf2freq_amd/libcnf2hip_x_$v.so; fi
    CNF2HIP_LIB=$lib timeout -k 10 300 python tools/iter_timing.py 2500 2500 2 2 ${AB_SF:-0.013} > gpurun_out/abacc_$v.log 2>&1 || { echo "$v failed"; tail -3 gpurun_out/abacc_$v.log; continue; }
    echo "$v: $(grep 'iteration 1' gpurun_out/abacc_$v.log | cut -c1-150)"
done
