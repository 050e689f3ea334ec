#!/bin/bash
# TEST INFRASTRUCTURE: builds tests/emu/libmentflow_emu.so — the csrc kernels compiled for the HOST against the
# fiber emulator in hip_emu.h (see that header).  Never loaded by the product package.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/../../mentflow_amd/csrc"
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
FLAGS="-std=c++17 -O1 -g -fPIC -DMF_EMU -include $HERE/hip_emu.h -Wno-unknown-attributes -Wno-unused-value -ffp-contract=off"
for f in api kde flow; do
  $CXX $FLAGS -x c++ -c "$SRC/$f.hip" -o "$HERE/$f.emu.o" &
done
$CXX -std=c++17 -O1 -g -fPIC -DMF_EMU -c "$HERE/hip_emu.cpp" -o "$HERE/hip_emu.emu.o" &
wait
$CXX -shared -o "$HERE/libmentflow_emu.so" "$HERE"/*.emu.o
echo "built $HERE/libmentflow_emu.so"
