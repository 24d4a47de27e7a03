#!/bin/bash
# TEST TOOL: the host engine's threads under ThreadSanitizer.  Builds the emulator library with csrc/engine.hip TSan-instrumented
# and a C++ stress harness on the C ABI (tools/emu_tsan/stress.cpp: concurrent generate() calls of several sessions through the
# decode pool, each result compared with the lone call), and runs it WITHOUT Python in the process.
# usage: tools/emu_tsan.sh [sessions] [calls per session]      (round 5, with the pool's hold policy: 4 x 3 calls, no report)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
OUT=${VC_TSAN_DIR:-/tmp/vcoder_tsan}
mkdir -p "$OUT"
cd "$ROOT"
python tools/emu_tsan/prepare.py "$OUT/inputs.txt"
cd "$ROOT/tests/emu"
./build_emu.sh > /dev/null
SRC=../../vcoder_amd/csrc
$CXX -x c++ -std=c++17 -O1 -g -fPIC -DVC_EMU -fsanitize=thread -I. -I$SRC -c $SRC/engine.hip -o "$OUT/engine.o"
OBJS=$(ls build/*.o | grep -v "/engine.o\|/kernel_api.o" | tr '\n' ' ')
# the emulator runtime itself stays uninstrumented: TSan's shadow call stack does not survive its hand-rolled fiber switches
$CXX -std=c++17 -O2 -fPIC -DVC_EMU -I. -I$SRC -c emu_runtime.cpp -o "$OUT/emu_runtime.o"
$CXX -std=c++17 -O2 -fPIC -DVC_EMU -fsanitize=thread -I. -I$SRC -shared "$OUT/emu_runtime.o" build/kernel_api.o $OBJS "$OUT/engine.o" \
    -o "$OUT/libvcoder_emu.so" -lpthread -ldl
$CXX -std=c++17 -O1 -g -fsanitize=thread -I"$ROOT/include" "$ROOT/tools/emu_tsan/stress.cpp" -o "$OUT/stress" -L"$OUT" -lvcoder_emu \
    -Wl,-rpath,"$OUT" -lpthread
cd "$ROOT"
TSAN_OPTIONS="halt_on_error=0:second_deadlock_stack=1" "$OUT/stress" "$OUT/inputs.txt" ${1:-4} ${2:-3}
