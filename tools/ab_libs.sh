#!/bin/bash
# A/B timing of library builds (tuning aid, GPU box): bash tools/ab_libs.sh name1 name2 ...  -> kernel_ms of the headline
# bench with cnf2freq_amd/libcnf2hip_x_<name>.so in place of the product library ("base" = the product library)
for v in "$@"; do
    if [ "$v" = base ]; then lib=$PWD/cnf2freq_amd/libcnf2hip.so; else lib=$PWD/cnf2freq_amd/libcnf2hip_x_$v.so; fi
    CNF2HIP_LIB=$lib timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-iteration-probe --no-merge-probe ${AB_FLAGS} \
        > gpurun_out/ab_$v.log 2>&1 || { echo "$v failed"; tail -3 gpurun_out/ab_$v.log; continue; }
    python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
r = json.loads(open("gpurun_out/ab_%s.log" % v).read().strip().split("\n")[-1])
print("%-12s kernel_ms %.1f  frac %.4f  checks %s" % (v, r["roofline"]["kernel_ms"], r["roofline"]["frac"], all(r["checks"].values())))
PY
done
