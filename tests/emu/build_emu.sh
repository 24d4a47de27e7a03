#!/bin/bash
# TEST-ONLY: builds the CPU emulator of the kernels (tests/emu/libvcoder_emu.so) with the host clang.
# `build_emu.sh f16`: the -DVC_F16 build of the same sources (IEEE fp16 MFMA operands, vc_device.h) -> libvcoder_emu_f16.so
set -e
cd "$(dirname "$0")"
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
SRC=../../vcoder_amd/csrc
OBJS=""
B=build; OUT=libvcoder_emu.so; DEF=""
if [ "$1" = "f16" ]; then B=build_f16; OUT=libvcoder_emu_f16.so; DEF="-DVC_F16"; fi
for f in $SRC/gemm.hip $SRC/norm.hip $SRC/attn.hip $SRC/decode.hip $SRC/misc.hip $SRC/select.hip $SRC/strict.hip $SRC/preprocess.hip $SRC/engine.hip $SRC/comm.hip; do
  [ -f "$f" ] || continue
  o=$B/$(basename $f .hip).o
  mkdir -p $B
  stale=0
  for d in "$f" $SRC/vc_device.h $SRC/kernels.h hip_emu.h $SRC/engine_*.inc; do [ "$d" -nt "$o" ] && stale=1; done
  if [ ! -f "$o" ] || [ $stale = 1 ]; then
    $CXX -x c++ -std=c++17 -O2 -fPIC -DVC_EMU $DEF -I. -I$SRC -c "$f" -o "$o"
  fi
  OBJS="$OBJS $o"
done
$CXX -std=c++17 -O2 -fPIC -DVC_EMU $DEF -I. -I$SRC -c $SRC/kernel_api.cpp -o $B/kernel_api.o
$CXX -std=c++17 -O2 -fPIC -DVC_EMU $DEF -I. -I$SRC -shared emu_runtime.cpp $B/kernel_api.o $OBJS -o $OUT -lpthread -ldl
echo built tests/emu/$OUT
