#!/bin/bash
# A/B of library builds on the iteration probe of bench.py (tuning aid, GPU box), one call, one box:
#   bash tools/ab_probe.sh name1 name2 ...   ("base" = the product library, else cnf2freq_amd/libcnf2hip_x_<name>.so)
# AB_KSTATS=1 adds the kernel table of each variant (rocprofv3 --kernel-trace --stats)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
    if [ "$v" = base ]; then lib=$R/cnf2freq_amd/libcnf2hip.so; else lib=$R/cnf2freq_amd/libcnf2hip_x_$v.so; fi
    # the host library binds the libcnf2hip.so of its own directory: a directory per variant
    d=/tmp/ab_probe_$v; rm -rf $d; mkdir -p $d; cp $lib $d/libcnf2hip.so; cp $R/cnf2freq_amd/libcnf2host.so $d/
    lib=$d/libcnf2hip.so; export CNF2HOST_LIB=$d/libcnf2host.so
    if [ -n "$AB_KSTATS" ]; then
        rm -rf $R/gpurun_out/ab_probe_$v
        CNF2HIP_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_probe_$v -- \
            python3 $R/tools/probe_iterations.py 500 2500 4 2 5 > $R/gpurun_out/ab_probe_$v.log 2>&1 || { echo "$v failed"; tail -3 $R/gpurun_out/ab_probe_$v.log; continue; }
    else
        CNF2HIP_LIB=$lib timeout -k 10 300 python3 $R/tools/probe_iterations.py 500 2500 4 2 5 > $R/gpurun_out/ab_probe_$v.log 2>&1 || { echo "$v failed"; tail -3 $R/gpurun_out/ab_probe_$v.log; continue; }
    fi
    echo "== $v: $(grep -E '"(iteration_s|update_s|sweep_accumulate_s)"' $R/gpurun_out/ab_probe_$v.log | tr -d '\n ')"
    if [ -n "$AB_KSTATS" ]; then
        f=$(find $R/gpurun_out/ab_probe_$v -name "*kernel_stats.csv" | head -1)
        python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print("   ", r["Name"][:60].ljust(60), r["Calls"].rjust(4), "total ms %8.1f" % (float(r["TotalDurationNs"]) / 1e6))
PY
    fi
done
