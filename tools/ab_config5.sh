R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
    if [ "$v" = base ]; then lib=$R/cnf2freq_amd/libcnf2hip.so; else lib=$R/cnf2freq_amd/libcnf2hip_x_$v.so; fi
    d=/tmp/ab_c5_$v; rm -rf $d; mkdir -p $d; cp $lib $d/libcnf2hip.so; cp $R/cnf2freq_amd/libcnf2host.so $d/
    export CNF2HIP_LIB=$d/libcnf2hip.so CNF2HOST_LIB=$d/libcnf2host.so
    timeout -k 10 500 python3 $R/tools/run_config5.py 2500 2500 4 100 400 > $d/run.log 2>&1
    echo "== $v: $(tail -1 $d/run.log | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["iterations_done"], "iterations", round(j["iterations_total_s"],1), "s, mean", round(j["iteration_s_mean"],4))')"
done
