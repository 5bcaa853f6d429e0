cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/gap_prof
timeout -k 10 500 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gap_prof -o gap -- python3 $GRAFT_REPO_ROOT/tools/run_config5.py 2500 2500 4 5 > $GRAFT_REPO_ROOT/gpurun_out/gap_run.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/gap_trace.py $GRAFT_REPO_ROOT/gpurun_out/gap_prof 0.25 > $GRAFT_REPO_ROOT/gpurun_out/gap_report.txt 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/gap_report.txt
rm -rf $GRAFT_REPO_ROOT/gpurun_out/gap_prof
