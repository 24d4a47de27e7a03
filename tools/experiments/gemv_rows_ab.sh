#!/bin/bash
# A/B timing of the 32-row decode GEMV (results of the variants are garbage, only launch times mean anything):
#   build:  tools/experiments/gemv_rows_ab.sh build     -> tools/experiments/lib/libvcoder_hip_dbg{1,2}.so
#           dbg1 = 32-row MFMA / LDS-read work on ONE fetched activation piece (no extra L2 -> LDS traffic)
#           dbg2 = all four activation pieces fetched, ONE MFMA row group
#   run (on the GPU box, inside a scratch copy): tools/experiments/gemv_rows_ab.sh run
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
OUT="$ROOT/tools/experiments/lib"
if [ "$1" = build ]; then
    mkdir -p "$OUT"
    python -m vcoder_amd.build > /dev/null
    for v in 1 2; do
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -DVC_GEMV_DBG=$v \
            -x hip -c "$ROOT/vcoder_amd/csrc/decode.hip" -o "$OUT/decode_dbg$v.o"
        objs=$(ls "$ROOT"/vcoder_amd/lib/obj/*.o | grep -v "/decode.o")
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs "$OUT/decode_dbg$v.o" -o "$OUT/libvcoder_hip_dbg$v.so"
    done
    ls -la "$OUT"
else
    cd "$ROOT"
    cp vcoder_amd/lib/libvcoder_hip.so /tmp/libvcoder_hip_real.so
    echo "== real"; python tools/kbench.py gemv_rows 2>&1 | grep "all GEMVs"
    for v in 1 2; do
        cp "$OUT/libvcoder_hip_dbg$v.so" vcoder_amd/lib/libvcoder_hip.so
        echo "== dbg$v"; python tools/kbench.py gemv_rows 2>&1 | grep "all GEMVs"
    done
    cp /tmp/libvcoder_hip_real.so vcoder_amd/lib/libvcoder_hip.so
fi
