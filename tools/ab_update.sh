#!/bin/bash
# A/B timing of library builds on the update pass (tuning aid, GPU box): bash tools/ab_update.sh name1 name2 ...
# -> kernel times of tools/iter_timing.py (500 families, 2 iterations from the initial state, scale factor 0.19) with
# cnf2freq_amd/libcnf2hip_x_<name>.so in place of the product library ("base" = the product library)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
    if [ "$v" = base ]; then lib=$R/cnf2freq_amd/libcnf2hip.so; else lib=$R/cnf2freq_amd/libcnf2hip_x_$v.so; fi
    rm -rf $R/gpurun_out/ab_upd_$v
    CNF2HIP_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_upd_$v -- \
        python3 $R/tools/iter_timing.py 500 2500 4 2 0.19 > $R/gpurun_out/ab_upd_$v.log 2>&1 || { echo "$v failed"; tail -3 $R/gpurun_out/ab_upd_$v.log; continue; }
    f=$(find $R/gpurun_out/ab_upd_$v -name "*kernel_stats.csv" | head -1)
    echo "== $v"; grep "^iteration" $R/gpurun_out/ab_upd_$v.log | cut -c1-120
    python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print("   ", r["Name"][:42].ljust(42), r["Calls"], "avg ms %.2f" % (float(r["AverageNs"]) / 1e6))
PY
done
