#!/bin/bash
# A/B of library builds on the three bench workloads (tuning aid, GPU box):
#   bash tools/ab_shapes.sh "f2 ail outbred" base prev NOPRODUCE      -> kernel_ms per (workload, library)
#   AB_EXTRA="--extra-flags 8" adds flags to every run (8 = CNF2_NO_TIES, 262144 = CNF2_STATIC_JOBS)
wl="$1"; shift
for w in $wl; do
    for v in "$@"; do
        if [ "$v" = base ]; then lib=$PWD/cnf2freq_amd/libcnf2hip.so; else lib=$PWD/cnf2freq_amd/libcnf2hip_x_$v.so; fi
        log=gpurun_out/abs_${w}_${v}.log
        CNF2HIP_LIB=$lib timeout -k 10 240 python bench.py --workload $w --steps 3 --warmup 1 --cpu-seconds 0 --no-iteration-probe --no-merge-probe ${AB_EXTRA} \
            > $log 2>&1 || { echo "$w $v failed"; tail -3 $log; continue; }
        python - "$w" "$v" "$log" "${AB_EXTRA}" <<'PY'
import json, sys
w, v, log, extra = sys.argv[1:5]
r = json.loads(open(log).read().strip().split("\n")[-1])
print("%-8s %-10s %-22s kernel_ms %8.2f  frac %.4f  clock %.0f MHz  checks %s" % (w, v, extra, r["roofline"]["kernel_ms"], r["roofline"]["frac"],
      r["roofline"].get("sweep_kernel_clock_mhz") or 0, all(r["checks"].values())), flush=True)
PY
    done
done
