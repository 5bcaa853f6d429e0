#!/bin/bash
# A/B of library builds over a long run (tuning aid, GPU box): the iteration probe (early regime) and the mean iteration time of
# iterations 51 - 70 of a 1 500-family run (late regime: step size ~0.16, four flows in five end in the scouts).
#   bash tools/ab_late.sh name1 name2 ...   ("base" = the product library, else cnf2freq_amd/libcnf2hip_x_<name>.so)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
    unset CNF2_AB_FLAGS
    if [ "$v" = base ]; then lib=$R/cnf2freq_amd/libcnf2hip.so;
    elif [ "$v" = literal ]; then lib=$R/cnf2freq_amd/libcnf2hip.so; export CNF2_AB_FLAGS=524288;      # CNF2_UPDATE_LITERAL_FINISH (the late part only)
    else lib=$R/cnf2freq_amd/libcnf2hip_x_$v.so; fi
    d=/tmp/ab_late_$v; rm -rf $d; mkdir -p $d; cp $lib $d/libcnf2hip.so; cp $R/cnf2freq_amd/libcnf2host.so $d/
    export CNF2HIP_LIB=$d/libcnf2hip.so CNF2HOST_LIB=$d/libcnf2host.so CNF2_NO_STATS=1
    timeout -k 10 300 python3 $R/tools/probe_iterations.py 500 2500 4 2 5 > $d/probe.log 2>&1
    timeout -k 10 400 python3 $R/tools/flow_stats.py ${AB_FAMS:-1500} 2500 4 70 > $d/late.log 2>&1
    echo "== $v: probe $(grep -E '"(iteration_s|update_s)"' $d/probe.log | tr -d '\n ') late iterations 51-70 mean $(grep '^it ' $d/late.log | awk '$2 > 50 {s += $3; n++} END {printf "%.4f s", s / n}')"
done
