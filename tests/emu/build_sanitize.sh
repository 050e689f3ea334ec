#!/bin/bash
# TEST INFRASTRUCTURE: builds tests/emu/sanitize_emu — the csrc kernels compiled for the HOST against the fiber
# emulator with AddressSanitizer + UndefinedBehaviorSanitizer, linked with the driver sanitize_main.cpp (SURVEY.md §5:
# "sanitizers on the host build").  GPU-side sanitizers are not available on this pool; this is the CPU pass.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/../../mentflow_amd/csrc"
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
# ASan through calls (-asan-instrumentation-with-call-threshold=0): inline instrumentation of the huge unrolled kernel
# bodies takes > 20 minutes to compile, the call form 1.5
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -mllvm -asan-instrumentation-with-call-threshold=0"
FLAGS="-std=c++17 -O1 -g -fPIC -DMF_EMU -include $HERE/hip_emu.h -Wno-unknown-attributes -Wno-unused-value -Wno-pass-failed -ffp-contract=off $SAN"
mkdir -p "$HERE/san"
# flow*.hip (huge unrolled bodies) get ASan + the cheap UBSan checks; the full UBSan set on it takes > 15 minutes to
# compile.  kde.hip, api.hip, the emulator and the driver get the full set.
UB_LIGHT="-fsanitize=address,bounds,shift,integer-divide-by-zero,unreachable,return,bool,enum,vla-bound -fno-sanitize-recover=all -fno-omit-frame-pointer -mllvm -asan-instrumentation-with-call-threshold=0"
FLAGS_LIGHT="-std=c++17 -O1 -g -fPIC -DMF_EMU -include $HERE/hip_emu.h -Wno-unknown-attributes -Wno-unused-value -Wno-pass-failed -ffp-contract=off $UB_LIGHT"
rm -f "$HERE"/san/*.o
while read -r name src flags; do          # one object per line of SOURCES.txt
  case "$name" in ""|\#*) continue;; esac
  case "$name" in
    flow*) $CXX $FLAGS_LIGHT $flags -x c++ -c "$SRC/$src" -o "$HERE/san/$name.o" & ;;
    *)     $CXX $FLAGS $flags -x c++ -c "$SRC/$src" -o "$HERE/san/$name.o" & ;;
  esac
done < "$SRC/SOURCES.txt"
$CXX -std=c++17 -O1 -g -fPIC -DMF_EMU $SAN -c "$HERE/hip_emu.cpp" -o "$HERE/san/hip_emu.o" &
$CXX $FLAGS -x c++ -c "$HERE/sanitize_main.cpp" -o "$HERE/san/main.o" &
wait
$CXX -fsanitize=address,undefined -o "$HERE/sanitize_emu" "$HERE"/san/*.o
echo "built $HERE/sanitize_emu"
