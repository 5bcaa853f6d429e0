#!/bin/bash
# Timing ablations of the sweep kernel (tuning aid): each variant library is built with one part of the
# kernel compiled out (-DCNF2_X_*), so its results are wrong and only kernel_ms means anything.
# usage (GPU box): bash tools/ablate.sh NOLOAD NOSTORE NOMEM
for v in "$@"; do
    CNF2HIP_LIB=$PWD/cnf2freq_amd/libcnf2hip_x_$v.so timeout -k 10 150 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-iteration-probe \
        > gpurun_out/ablate_$v.log 2>&1 || { echo "$v failed"; tail -3 gpurun_out/ablate_$v.log; continue; }
    python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
r = json.loads(open("gpurun_out/ablate_%s.log" % v).read().strip().split("\n")[-1])
print("%-10s ms_per_step %.1f kernel_ms %.1f" % (v, r["ms_per_step"], r["roofline"]["kernel_ms"]))
PY
done
