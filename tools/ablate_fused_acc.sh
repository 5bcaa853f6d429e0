#!/bin/bash
# Timing ablation (GPU box): what would forming the contractions v, u, z of HOT LOOP 2 in the sweep's backward pass buy?
# Builds the library with -DCNF2_X_FUSEDACC -- the accumulate instantiation of the sweep forms the three contractions (with
# stand-in tables for the HOMOZYGOUS probe sums) and writes 160 doubles per unit instead of the 512 posterior weights,
# acc_tile_kernel reads them and skips its phase B; results are WRONG, only the times mean anything -- and times
# sweep + accumulators against the product library at the iteration probe's and at config 5's size.
#   bash tools/ablate_fused_acc.sh   -> gpurun_out/ablate_fused_acc.log
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R/cnf2freq_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DCNF2_X_FUSEDACC -shared \
    -o ../libcnf2hip_x_FUSEDACC.so cnf2_kernels.hip cnf2_update_kernels.hip cnf2_capi.hip cnf2_window.cpp 2>/dev/null || { echo "build failed"; exit 1; }
cd $R
for size in "500 2500 4 2000" "2500 2500 4 10000"; do
    python tools/acc_timing.py $size 2>&1 | tail -1
    CNF2HIP_LIB=$R/cnf2freq_amd/libcnf2hip_x_FUSEDACC.so python tools/acc_timing.py $size 2>&1 | tail -1
done | tee gpurun_out/ablate_fused_acc.log
