#!/bin/bash
# The sanitizer pass of DESIGN.md section 5 ("Round 5: sanitizers over the device source"), CPU only:
#   tools/sanitize_cpu.sh [address|undefined] [pytest arguments ...]
# Builds the lane emulation (tests/emu/emu_kernel.cpp = the DEVICE headers compiled for the host) and the oracle with
# -fsanitize=<kind>, runs the CPU suite with the sanitizer's runtime preloaded into python, prints the reports it left
# and puts the ordinary builds back.  (GPU AddressSanitizer is not available on the pool; a lane of the emulation is in
# the states a GPU lane is in, so out-of-range table reads of inactive lanes show up here.)
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
KIND=${1:-address}; shift || true
ARGS=${@:-tests -q -m "not gpu"}
EMU=$R/tests/emu/libemu_kernel.so; ORC=$R/oracle/libmc_oracle.so
T=$(mktemp -d)
cp -p "$EMU" "$T/emu.so" 2>/dev/null; cp -p "$ORC" "$T/orc.so" 2>/dev/null
restore() {
  [ -f "$T/emu.so" ] && cp -p "$T/emu.so" "$EMU"; [ -f "$T/orc.so" ] && cp -p "$T/orc.so" "$ORC"
  touch "$EMU" "$ORC"; rm -rf "$T"
}
trap restore EXIT
FMA=""; grep -q fma /proc/cpuinfo && FMA="-mfma"
REC=""; [ "$KIND" = address ] && REC="-fsanitize-recover=address"
g++ -O1 -g -std=c++17 -fPIC -shared -ffp-contract=fast $FMA -fsanitize=$KIND $REC -fno-omit-frame-pointer -o "$EMU" "$R/tests/emu/emu_kernel.cpp" || exit 1
gcc -O1 -g -fPIC -shared -fopenmp -ffp-contract=off -fsanitize=$KIND $REC -fno-omit-frame-pointer -o "$ORC" "$R/oracle/mc_oracle.c" -lm || exit 1
LIBSAN=$(gcc -print-file-name=lib$([ "$KIND" = address ] && echo asan || echo ubsan).so)
cd "$R"
rm -f "$T"/san.*
LD_PRELOAD=$LIBSAN ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:log_path=$T/san UBSAN_OPTIONS=print_stacktrace=1:log_path=$T/san \
  eval python -m pytest $ARGS
RC=$?
N=$(ls "$T"/san.* 2>/dev/null | wc -l)
echo "sanitizer ($KIND): $N report file(s), pytest exit $RC"
for f in "$T"/san.*; do [ -f "$f" ] && { grep -m3 -E "ERROR|runtime error|SUMMARY" "$f"; }; done
[ "$N" = 0 ] && [ "$RC" = 0 ]
