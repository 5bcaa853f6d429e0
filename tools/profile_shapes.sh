#!/bin/bash
# rocprofv3 kernel stats for the BASELINE configurations other than the headline (run on the GPU box):
#   bash tools/profile_shapes.sh r02_a
#   config 3 (advanced intercross, tied windows) and config 5's shape (outbred, 20 % missing) at full size through
#   tools/shape_timing.py --full  -> gpurun_out/prof_<tag>/<tag>_kernel_stats_shapes_c3_c5.csv + the script's own log
# Copy what is to be judged into profiles/.
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/shapes -- \
    python3 $R/tools/shape_timing.py --full > $out/${tag}_shape_timing.log 2>&1 || echo "shape pass failed"
python3 - "$out" "$tag" <<'PY'
import csv, glob, sys
out, tag = sys.argv[1], sys.argv[2]
for f in glob.glob(out + "/shapes/**/*kernel_stats.csv", recursive=True):
    lines = open(f).read().split("\n")
    open("%s/%s_kernel_stats_shapes_c3_c5.csv" % (out, tag), "w").write("\n".join(lines[:12]) + "\n")
with open("%s/%s_kernel_trace_shapes_c3_c5.csv" % (out, tag), "w") as g:
    g.write("Kernel_Name,Duration_ms,Grid_Size,Workgroup_Size,LDS_Block_Size,Scratch_Size,VGPR_Count\n")
    for f in glob.glob(out + "/shapes/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fb_" in r["Kernel_Name"]:
                g.write("%s,%.3f,%s,%s,%s,%s,%s\n" % (r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6,
                        r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")),
                        r.get("LDS_Block_Size", ""), r.get("Scratch_Size", ""), r.get("VGPR_Count", "")))
PY
cat $out/${tag}_shape_timing.log | tail -5
