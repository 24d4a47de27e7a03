#!/bin/bash
# TEST-ONLY: builds the CPU emulator of the kernels (tests/emu/libvcoder_emu.so) with the host clang.
set -e
cd "$(dirname "$0")"
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
SRC=../../vcoder_amd/csrc
OBJS=""
for f in $SRC/gemm.hip $SRC/norm.hip $SRC/attn.hip $SRC/decode.hip $SRC/misc.hip $SRC/select.hip $SRC/strict.hip $SRC/preprocess.hip $SRC/engine.hip $SRC/comm.hip; do
  [ -f "$f" ] || continue
  o=build/$(basename $f .hip).o
  mkdir -p build
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ $SRC/vc_device.h -nt "$o" ] || [ $SRC/kernels.h -nt "$o" ] || [ hip_emu.h -nt "$o" ]; then
    $CXX -x c++ -std=c++17 -O2 -fPIC -DVC_EMU -I. -I$SRC -c "$f" -o "$o"
  fi
  OBJS="$OBJS $o"
done
$CXX -std=c++17 -O2 -fPIC -DVC_EMU -I. -I$SRC -c $SRC/kernel_api.cpp -o build/kernel_api.o
$CXX -std=c++17 -O2 -fPIC -DVC_EMU -I. -I$SRC -shared emu_runtime.cpp build/kernel_api.o $OBJS -o libvcoder_emu.so -lpthread -ldl
echo built tests/emu/libvcoder_emu.so
