#!/bin/bash
# The host C++ of the drop-in under sanitizers (CPU box):
#   bash tools/sanitize_host.sh [log]      -> profiles/r05_sanitizers_host.log by default
# 1. tests/sanitize/sanitize_host.cpp (readers, digit former, text, partition plan, update math, the shared-memory
#    transport with its ranks as threads) built with -fsanitize=address,undefined and with -fsanitize=thread;
# 2. the executable's reader path (`cnF2freq --parse-only`) built with -fsanitize=address,undefined and run by
#    tests/test_cli_readers.py (CNF2_EXE names the build).
# Zero reports = every run ends with its own "ok" and no line of a sanitizer in the log.
R=$(cd "$(dirname "$0")/.." && pwd)
LOG=${1:-$R/profiles/r05_sanitizers_host.log}
T=$(mktemp -d /tmp/cnf2san.XXXXXX)
trap 'rm -rf "$T"' EXIT
C=$R/cnf2freq_amd/csrc
INC="-I$C -I$R/include"
{
echo "== sanitize_host: g++ $(g++ -dumpversion), $(date -u +%Y-%m-%dT%H:%MZ)"
for san in address,undefined thread; do
    echo "-- -fsanitize=$san"
    g++ -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=$san -fno-sanitize-recover=all $INC -o $T/san_$san \
        $R/tests/sanitize/sanitize_host.cpp $C/host/cnf2_readers.cpp -lpthread || { echo "BUILD FAILED"; continue; }
    mkdir -p $T/d_$san
    ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 TSAN_OPTIONS=halt_on_error=0 $T/san_$san $T/d_$san
    echo "exit code $?"
done
echo "-- cnF2freq --parse-only with -fsanitize=address,undefined through tests/test_cli_readers.py"
ROCM=${ROCM:-/opt/rocm}
g++ -O1 -g -std=c++17 -fopenmp -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=all -D__HIP_PLATFORM_AMD__ -I$R/include -I$ROCM/include \
    -o $T/cnF2freq_asan $C/host/cnf2freq_main.cpp $C/host/cnf2_readers.cpp $C/host/cnf2_engine.cpp -L$R/cnf2freq_amd -lcnf2hip -L$ROCM/lib -lrccl -lamdhip64 \
    -lpthread -Wl,-rpath,$R/cnf2freq_amd -Wl,-rpath,$ROCM/lib \
    && (cd $R && ASAN_OPTIONS=detect_leaks=0 CNF2_EXE=$T/cnF2freq_asan python -m pytest tests/test_cli_readers.py -x -q 2>&1 | tail -3)
} 2>&1 | tee $LOG
if grep -q "ERROR: \|runtime error\|WARNING: ThreadSanitizer\|FAILED\|BUILD FAILED" $LOG; then echo "sanitize_host.sh: REPORTS FOUND"; exit 1; fi
echo "sanitize_host.sh: zero reports" | tee -a $LOG
