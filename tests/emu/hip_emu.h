// hip_emu.h — TEST-ONLY functional emulator for the gfx950 kernels (thread-per-lane, pthread barriers).
//
// Purpose: there is no GPU in the build container and only ~90 GPU-minutes per round, so the kernels'
// index arithmetic, LDS layouts, MFMA fragment plumbing and reductions are first checked on the CPU
// under the ASSUMED hardware semantics (wave64, v_mfma_f32_16x16x32_bf16 fragment maps).  The real
// parity tests (`-m gpu`) then only have to confirm the hardware semantics.  Nothing under
// vcoder_amd/ (the product) includes or links this file without -DVC_EMU, and the product library is
// built exclusively by hipcc.
//
// Blocks of one launch run sequentially (so `__shared__` can be a function-local static); the threads
// of a block are real OS threads synchronised by pthread barriers.
#pragma once
#include <pthread.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
#include <functional>
#include <thread>
#include <vector>
#include <algorithm>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;
using std::min;
using std::max;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__ __restrict

namespace vc_emu {
struct WaveCtx {
    pthread_barrier_t bar;
    alignas(16) unsigned char xchg[64][64];
};
struct BlockCtx {
    pthread_barrier_t bar;
    int nthreads;
    WaveCtx waves[16];
    char* dyn_smem;
};
extern thread_local BlockCtx* g_ctx;
extern thread_local int g_lane, g_wave;
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
}  // namespace vc_emu


// ---- minimal HIP runtime shims so that the host engine (engine.hip) also runs under the emulator ----------
#include <chrono>
typedef int hipError_t;
static const hipError_t hipSuccess = 0;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal };
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = aligned_alloc(256, (n + 255) / 256 * 256 + 256); return *p ? 0 : 1; }
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { free(p); return 0; }
inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return 0; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return 0; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return 0; }
inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
    for (size_t r = 0; r < h; ++r) memcpy((char*)d + r * dp, (const char*)s + r * sp, w);
    return 0;
}
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return 0; }
inline hipError_t hipSetDevice(int) { return 0; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = (void*)0x1; return 0; }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
struct EmuEvent { double t; };
typedef EmuEvent* hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new EmuEvent{0}; return 0; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return 0; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
    e->t = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    return 0;
}
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return 0; }
namespace vc_emu {
struct Graph { std::vector<std::function<void()>> ops; };
extern Graph* g_capturing;
}
typedef vc_emu::Graph* hipGraph_t;
typedef vc_emu::Graph* hipGraphExec_t;
typedef void* hipGraphNode_t;
inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { vc_emu::g_capturing = new vc_emu::Graph(); return 0; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = vc_emu::g_capturing; vc_emu::g_capturing = nullptr; return 0; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, hipGraphNode_t*, char*, size_t) { *e = new vc_emu::Graph(*g); return 0; }
inline hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return 0; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t g) { delete g; return 0; }
inline hipError_t hipGraphLaunch(hipGraphExec_t g, hipStream_t) { for (auto& f : g->ops) f(); return 0; }

extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

inline void __syncthreads() { pthread_barrier_wait(&vc_emu::g_ctx->bar); }
inline float __expf(float x) { return expf(x); }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __fmul_rn(float a, float b) { return a * b; }
inline float __fadd_rn(float a, float b) { return a + b; }

#define VC_DYNAMIC_SMEM(type, name) type* name = reinterpret_cast<type*>(vc_emu::g_ctx->dyn_smem)
#define VC_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    vc_emu::launch(grid, block, shmem, [=]() { kernel(__VA_ARGS__); })

namespace vc {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

inline int lane_id() { return vc_emu::g_lane; }

template <class T> inline T shfl_xor(T v, int mask) {
    static_assert(sizeof(T) <= 64, "xchg slot");
    auto& w = vc_emu::g_ctx->waves[vc_emu::g_wave];
    memcpy(w.xchg[vc_emu::g_lane], &v, sizeof(T));
    pthread_barrier_wait(&w.bar);
    T r;
    memcpy(&r, w.xchg[(vc_emu::g_lane ^ mask) & 63], sizeof(T));
    pthread_barrier_wait(&w.bar);
    return r;
}
template <class T> inline T shfl(T v, int src) {
    auto& w = vc_emu::g_ctx->waves[vc_emu::g_wave];
    memcpy(w.xchg[vc_emu::g_lane], &v, sizeof(T));
    pthread_barrier_wait(&w.bar);
    T r;
    memcpy(&r, w.xchg[src & 63], sizeof(T));
    pthread_barrier_wait(&w.bar);
    return r;
}

// v_mfma_f32_16x16x32_bf16 under the assumed fragment maps (see vc_device.h header).
inline f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
    auto& w = vc_emu::g_ctx->waves[vc_emu::g_wave];
    const int lane = vc_emu::g_lane;
    memcpy(w.xchg[lane], &a, 16);
    memcpy(w.xchg[lane] + 16, &b, 16);
    pthread_barrier_wait(&w.bar);
    const int j = lane & 15;
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int i = (lane >> 4) * 4 + r;
        float acc = 0.f;
        for (int k = 0; k < 32; ++k) {
            uint16_t ab, bb;
            memcpy(&ab, w.xchg[i + 16 * (k / 8)] + 2 * (k % 8), 2);
            memcpy(&bb, w.xchg[j + 16 * (k / 8)] + 16 + 2 * (k % 8), 2);
            uint32_t au = (uint32_t)ab << 16, bu = (uint32_t)bb << 16;
            float af, bf;
            memcpy(&af, &au, 4);
            memcpy(&bf, &bu, 4);
            acc += af * bf;
        }
        d[r] = c[r] + acc;
    }
    pthread_barrier_wait(&w.bar);
    return d;
}
}  // namespace vc
