#!/bin/sh
# Register / scratch / LDS use of the gfx950 kernels, from the code objects' metadata notes.
# usage: [CNF2_EXTRA_FLAGS="-D..."] tools/kernel_resources.sh [name filter]      (compiles the .hip files of cnf2freq_amd/csrc for the device only)
FILTER=${1:-.}
HERE=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d /tmp/cnf2res.XXXXXX)
trap 'rm -rf "$TMP"' EXIT
for f in "$HERE"/cnf2freq_amd/csrc/cnf2_kernels.hip "$HERE"/cnf2freq_amd/csrc/cnf2_update_kernels.hip; do
    b=$(basename "$f" .hip)
    (cd "$HERE/cnf2freq_amd/csrc" && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics --cuda-device-only \
        --no-gpu-bundle-output $CNF2_EXTRA_FLAGS -c -o "$TMP/$b.o" "$f" 2>/dev/null)
    /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$TMP/$b.o"
done | python3 -c '
import re, sys, subprocess
txt = sys.stdin.read()
flt = re.compile(sys.argv[1])
for blk in txt.split("- .agpr_count:")[1:]:
    get = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    dem = subprocess.run(["c++filt", get("name")], capture_output=True, text=True).stdout.strip()
    if not flt.search(dem): continue
    agpr = blk.split("\n")[0].strip()
    print("%-100s vgpr %3s agpr %3s sgpr %3s spill_v %3s scratch %4s B  lds %6s B" % (dem[:100], get("vgpr_count"), agpr, get("sgpr_count"), get("vgpr_spill_count"), get("private_segment_fixed_size"), get("group_segment_fixed_size")))
' "$FILTER"
