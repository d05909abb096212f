#!/bin/bash
# Builds tests/fuzz/fuzz_serial.cpp (libFuzzer + AddressSanitizer + UBSan, host only) and fuzzes the two byte parsers.
#   tests/fuzz/run.sh [seconds per worker = 300] [workers = 4] [work directory = a fresh temporary one]
# The committed seeds (tests/fuzz/corpus, written by make_corpus.py) are read-only inputs; new units go to the work directory.
# Exit status 0 = no crash, no sanitizer report, no hang.  -asan-globals=0: the ROCm build of the ASan runtime reports a
# bogus odr-violation for every instrumented global of a statically linked fuzzer binary and ignores detect_odr_violation=0.
set -eu
HERE=$(cd "$(dirname "$0")" && pwd)
SECS=${1:-300}; WORKERS=${2:-4}; WORK=${3:-$(mktemp -d /tmp/plonk_fuzz.XXXXXX)}
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
mkdir -p "$HERE/_build" "$WORK"
if [ ! -x "$HERE/_build/fuzz_serial" ] || [ "$HERE/fuzz_serial.cpp" -nt "$HERE/_build/fuzz_serial" ] || [ "$HERE/../../plonk_amd/csrc/serial_check.hpp" -nt "$HERE/_build/fuzz_serial" ]; then
  "$CXX" -std=c++17 -O1 -g -fsanitize=fuzzer,address,undefined -fno-sanitize-recover=undefined -mllvm -asan-globals=0 \
         "$HERE/fuzz_serial.cpp" -o "$HERE/_build/fuzz_serial"
fi
cd "$WORK"
if [ "$WORKERS" -gt 1 ]; then
  exec "$HERE/_build/fuzz_serial" -max_len=65536 -timeout=20 -rss_limit_mb=2048 -max_total_time="$SECS" -jobs="$WORKERS" -workers="$WORKERS" -print_final_stats=1 "$WORK" "$HERE/corpus"
else
  exec "$HERE/_build/fuzz_serial" -max_len=65536 -timeout=20 -rss_limit_mb=2048 -max_total_time="$SECS" -print_final_stats=1 "$WORK" "$HERE/corpus"
fi
