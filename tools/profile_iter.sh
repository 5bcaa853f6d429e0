#!/bin/bash
# rocprofv3 kernel stats of one full haplotyping iteration at BASELINE config 5's size (run on the GPU box):
#   bash tools/profile_iter.sh r02_c [scalefactor]  -> gpurun_out/prof_<tag>/<tag>_kernel_stats_config5_iteration.csv + timing log
tag=${1:-rXX}
sf=${2:-0.013}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/iter -- \
    python3 $R/tools/iter_timing.py 2500 2500 4 2 $sf > $out/${tag}_iter_timing_config5.log 2>&1 || echo "iteration pass failed"
python3 - "$out" "$tag" <<'PY'
import glob, sys
out, tag = sys.argv[1], sys.argv[2]
for f in glob.glob(out + "/iter/**/*kernel_stats.csv", recursive=True):
    lines = open(f).read().split("\n")
    open("%s/%s_kernel_stats_config5_iteration.csv" % (out, tag), "w").write("\n".join(lines[:14]) + "\n")
PY
grep -v "^[EW]2026" $out/${tag}_iter_timing_config5.log | tail -6
