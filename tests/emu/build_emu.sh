#!/bin/bash
# TEST INFRASTRUCTURE: builds tests/emu/libmentflow_emu.so — the csrc kernels compiled for the HOST against the
# fiber emulator in hip_emu.h (see that header).  Never loaded by the product package.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/../../mentflow_amd/csrc"
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
FLAGS="-std=c++17 -O1 -g -fPIC -DMF_EMU -include $HERE/hip_emu.h -Wno-unknown-attributes -Wno-unused-value -Wno-unused-function -ffp-contract=off"
rm -f "$HERE"/*.emu.o
while read -r name src flags; do          # one object per line of SOURCES.txt (not a pipeline: `wait` must see the jobs)
  case "$name" in ""|\#*) continue;; esac
  $CXX $FLAGS $flags -x c++ -c "$SRC/$src" -o "$HERE/$name.emu.o" &
done < "$SRC/SOURCES.txt"
$CXX -std=c++17 -O1 -g -fPIC -DMF_EMU -c "$HERE/hip_emu.cpp" -o "$HERE/hip_emu.emu.o" &
wait
$CXX -shared -o "$HERE/libmentflow_emu.so" "$HERE"/*.emu.o
echo "built $HERE/libmentflow_emu.so"
