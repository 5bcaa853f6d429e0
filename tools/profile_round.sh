#!/bin/bash
# rocprofv3 evidence for one kernel version (run on the GPU box):  bash tools/profile_round.sh r01_f
#   1. --kernel-trace --stats on the full-size bench         -> <tag>_kernel_stats.csv, <tag>_kernel_trace_fb_fast.csv
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes -> <tag>_hbm_traffic.json (gfx950 corrections applied)
#   3. SQ counter passes on a 2000-individual slice           -> <tag>_pmc_sq_summary_2000inds.txt
# Everything lands in gpurun_out/prof_<tag>/; copy what is to be judged into profiles/.
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- \
    python3 $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-iteration-probe > $out/stats.log 2>&1 || echo "stats pass failed"
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -- \
        python3 $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-iteration-probe > $out/$c.log 2>&1 || echo "$c pass failed"
done
python3 - "$out" "$tag" "$R" <<'PY'
import csv, glob, hashlib, json, os, sys
out, tag, root = sys.argv[1], sys.argv[2], sys.argv[3]
h = hashlib.sha256()          # same identity as bench.py kernel_source_sha()
for f in ("cnf2_kernels.hip", "cnf2_emtab.h", "cnf2_emission.h", "cnf2_lane.h", "cnf2_device.h"):
    h.update(open(os.path.join(root, "cnf2freq_amd", "csrc", f), "rb").read())
src_sha = h.hexdigest()[:16]
def rows(pattern):
    for f in glob.glob(out + pattern, recursive=True):
        yield from csv.DictReader(open(f))
# kernel stats (top rows) and the sweep kernel's trace rows
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    lines = open(f).read().split("\n")
    open("%s/%s_kernel_stats_full_f2_10k_x_50k.csv" % (out, tag), "w").write("\n".join(lines[:12]) + "\n")
tr = [r for r in rows("/stats/**/*kernel_trace.csv") if "fb_fast" in r["Kernel_Name"]]
with open("%s/%s_kernel_trace_fb_fast_kernel.csv" % (out, tag), "w") as g:
    g.write("Kernel_Name,Start_Timestamp,End_Timestamp,Duration_ms,Grid_Size,Workgroup_Size,LDS_Block_Size,Scratch_Size,VGPR_Count,SGPR_Count\n")
    for r in tr:
        g.write("%s,%s,%s,%.3f,%s,%s,%s,%s,%s,%s\n" % (r["Kernel_Name"], r["Start_Timestamp"], r["End_Timestamp"],
                (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size_X", r.get("Grid_Size", "")),
                r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("LDS_Block_Size", ""), r.get("Scratch_Size", ""),
                r.get("VGPR_Count", ""), r.get("SGPR_Count", "")))
avg_ms = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in tr) / max(1, len(tr))
def per_launch(counter):
    v = [float(r["Counter_Value"]) for r in rows("/%s/**/*counter_collection.csv" % counter)
         if "fb_fast" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    n = len([1 for r in rows("/%s/**/*kernel_trace.csv" % counter) if "fb_fast" in r["Kernel_Name"]]) or 1
    return sum(v) / n
fetch_kb, write_kb = per_launch("FETCH_SIZE"), per_launch("WRITE_SIZE")
inds, markers = 10000, 50020
fb, wb = fetch_kb * 1024 * 2, write_kb * 1024
json.dump({"inds": inds, "markers": markers, "kernel": "cnf2::fb_fast_kernel<true>", "kernel_src_sha": src_sha,
           "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
           "correction": "FETCH_SIZE x2 on gfx950 for wide coalesced reads; WRITE_SIZE exact (MI355X_MICROARCH.md section HBM); separate --pmc passes",
           "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "bytes_per_launch": fb + wb,
           "bytes_per_unit": (fb + wb) / (inds * markers), "algorithmic_bytes_per_unit": 8248,
           "rocprof_kernel_avg_ms": avg_ms,
           "source": "tools/profile_round.sh %s: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-iteration-probe" % tag},
          open("%s/%s_hbm_traffic.json" % (out, tag), "w"), indent=1)
print("kernel avg ms", avg_ms, "bytes/unit", (fb + wb) / (inds * markers))
PY
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/sq$i -- \
        python3 $R/bench.py --inds 2000 --steps 1 --warmup 0 --cpu-seconds 0 --no-iteration-probe > $out/sq$i.log 2>&1 || echo "sq pass $i failed"
done
python3 $R/profiles/pmc_summarize.py $out/sq1 $out/sq2 $out/sq3 $out/sq4 --units $((2000*50020)) --kernel fb_fast \
    > $out/${tag}_pmc_sq_summary_2000inds.txt
# VALU instructions per unit into the traffic file: bench.py turns them into valu_issue_frac with the clock it measures
python3 - "$out" "$tag" <<'PY'
import json, re, sys
out, tag = sys.argv[1], sys.argv[2]
txt = open("%s/%s_pmc_sq_summary_2000inds.txt" % (out, tag)).read()
m = re.search(r"SQ_INSTS_VALU\s+\S+ \(dispatch rows \d+\)\s+per unit (\S+)", txt)
f = "%s/%s_hbm_traffic.json" % (out, tag)
j = json.load(open(f))
j["valu_per_unit"] = float(m.group(1)) if m else None
json.dump(j, open(f, "w"), indent=1)
print("valu_per_unit", j["valu_per_unit"])
PY
cat $out/${tag}_pmc_sq_summary_2000inds.txt
ls $out
