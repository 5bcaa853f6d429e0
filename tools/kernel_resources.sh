#!/bin/sh
# Register / scratch / LDS use of every gfx950 kernel in libcnf2hip.so, from the code object's metadata notes.
# usage: tools/kernel_resources.sh [library] [name filter]
LIB=${1:-cnf2freq_amd/libcnf2hip.so}
FILTER=${2:-.}
LLVM=/opt/rocm/lib/llvm/bin
TMP=$(mktemp -d /tmp/cnf2res.XXXXXX)
trap 'rm -rf "$TMP"' EXIT
$LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$LIB" --output="$TMP/dev.co" --unbundle 2>/dev/null || {
    # the bundle sits in the .hip_fatbin section of a shared object
    $LLVM/llvm-objcopy -O binary --only-section=.hip_fatbin "$LIB" "$TMP/fat.bin"
    $LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$TMP/fat.bin" --output="$TMP/dev.co" --unbundle
}
$LLVM/llvm-readelf --notes "$TMP/dev.co" | python3 -c '
import re, sys
txt = sys.stdin.read()
flt = re.compile(sys.argv[1])
for blk in txt.split("- .agpr_count:")[1:]:
    get = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = get("name")
    import subprocess
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if not flt.search(dem): continue
    agpr = blk.split("\n")[0].strip()
    print("%-110s vgpr %3s agpr %3s sgpr %3s spill_v %3s scratch %4s B  lds %6s B" % (dem[:110], get("vgpr_count"), agpr, get("sgpr_count"), get("vgpr_spill_count"), get("private_segment_fixed_size"), get("group_segment_fixed_size")))
' "$FILTER"
