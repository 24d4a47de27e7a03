#!/bin/bash
# TEST TOOL: the host engine (csrc/engine.hip — splice planner, KV management, the decode pool's threads and queues) under
# AddressSanitizer.  Builds a variant of the emulator library in which engine.hip is ASan-instrumented (the kernels run on the
# emulator's own lane fibers and stay uninstrumented: their stack switching is not something ASan follows) and runs the engine /
# pool / fuzz / split tests against it.  usage: tools/emu_asan.sh [pytest args...]     (round 3: 48 tests, no report; round 5: 56 tests, no report)
# VC_SAN=ubsan: UndefinedBehaviorSanitizer (minimal runtime) instead — reports print as "ubsan: <kind>" lines (rounds 3 and 5: none).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
if [ "${VC_SAN:-asan}" = ubsan ]; then
    SANC="-fsanitize=undefined -fsanitize-minimal-runtime -fno-sanitize=vptr,function"; SANL="$SANC"; RTN=libclang_rt.ubsan_minimal-x86_64.so
else
    SANC="-fsanitize=address -fno-omit-frame-pointer"; SANL="-fsanitize=address -shared-libasan"; RTN=libclang_rt.asan-x86_64.so
fi
RT=$(dirname "$($CXX -print-file-name=libclang_rt.asan-x86_64.so)")/$RTN
OUT=${VC_ASAN_DIR:-/tmp/vcoder_asan}
mkdir -p "$OUT"
cd "$ROOT/tests/emu"
./build_emu.sh > /dev/null
SRC=../../vcoder_amd/csrc
$CXX -x c++ -std=c++17 -O1 -g -fPIC -DVC_EMU $SANC -I. -I$SRC -c $SRC/engine.hip -o "$OUT/engine.o"
OBJS=$(ls build/*.o | grep -v "/engine.o\|/kernel_api.o" | tr '\n' ' ')
$CXX -std=c++17 -O2 -fPIC -DVC_EMU $SANL -I. -I$SRC -shared emu_runtime.cpp build/kernel_api.o $OBJS "$OUT/engine.o" \
    -o "$OUT/libvcoder_emu.so" -lpthread -ldl
cd "$ROOT"
TESTS=${@:-tests/test_engine_emu.py tests/test_pool_emu.py tests/test_fuzz_emu.py tests/test_split_emu.py}
VC_EMU_LIB="$OUT/libvcoder_emu.so" LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
    python -m pytest $TESTS -x -q -s -p no:cacheprovider
