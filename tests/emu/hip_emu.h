// hip_emu.h — TEST-ONLY functional emulator for the gfx950 kernels (fiber-per-lane, cooperative scheduling).
//
// Purpose: there is no GPU in the build container and only ~90 GPU-minutes per round, so the kernels'
// index arithmetic, LDS layouts, MFMA fragment plumbing and reductions are first checked on the CPU
// under the ASSUMED hardware semantics (wave64, v_mfma_f32_16x16x32_bf16 fragment maps).  The real
// parity tests (`-m gpu`) then only have to confirm the hardware semantics.  Nothing under
// vcoder_amd/ (the product) includes or links this file without -DVC_EMU, and the product library is
// built exclusively by hipcc.
//
// Every GPU thread is a user-level fiber (hand-rolled x86-64 context switch); the fibers of one workgroup run
// round-robin on ONE OS thread and yield at barriers / cross-lane collectives, so `__shared__` is a
// `static thread_local` and several workgroups run concurrently on different OS threads.
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
#include <functional>
#include <thread>
#include <vector>
#include <algorithm>
#include <map>
#include <mutex>
#include <sys/mman.h>

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
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define __restrict__ __restrict

namespace vc_emu {
struct WaveCtx {
    int arrived = 0;
    unsigned gen = 0;
    alignas(16) unsigned char xchg[64][64];
    // LDS base of the wave's k-th DMA as the first lane to issue it passed it: the hardware takes the base from M0 (a scalar), so
    // every lane must pass the same one — a lane-dependent base would work here (each lane copies to its own) and not there
    const void* dma_base[256];
    unsigned dma_base_seq[256];
};
struct BlockCtx;
// A lane's LDS-DMAs in flight (global_load_lds): issued in program order, LANDED only when the lane waits for them — the
// vmcnt model below.  128 entries = far more than any kernel keeps outstanding.
struct PendingDma {
    const void* src;
    void* dst;
};
struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    bool done = false;
    int lane = 0, wave = 0;
    dim3 tidx;
    BlockCtx* blk = nullptr;
    PendingDma dma[128];
    unsigned dma_head = 0, dma_tail = 0;   // ring: [head, tail) pending, oldest first
    unsigned dma_seq = 0;                  // DMAs this lane has issued in the block (wave-uniform control flow: the same for all lanes)
    const void* lds_rd[64];                // dynamic-LDS addresses this lane read since its last lgkmcnt wait (ring, newest last)
    unsigned lds_rd_head = 0, lds_rd_tail = 0;
};
// One entry per 16 bytes of a launch's dynamic LDS (VC_EMU_RACE=1): who wrote / read it last and in which barrier epoch
struct LdsShadow {
    unsigned w_epoch, r_epoch;
    short w_wave, r_wave;   // -1: nobody yet; r_wave -2: several waves read it in r_epoch
};
struct BlockCtx {
    size_t dyn_bytes = 0;
    LdsShadow* shadow = nullptr;
    int nthreads = 0, alive = 0;
    int bar_arrived = 0;
    unsigned bar_gen = 0;
    int wave_alive[16];
    WaveCtx waves[16];
    char* dyn_smem = nullptr;
    dim3 bidx, bdim, gdim;
    unsigned long progress = 0;
    const std::function<void()>* body = nullptr;
    void* sched_sp = nullptr;
};
extern thread_local Fiber* g_cur;
void yield_to_scheduler();
// vmcnt model (VC_EMU_DMA=0: DMAs land at once, waits are no-ops): an LDS-DMA is queued at issue and its 16 bytes are copied
// when the issuing lane executes a wait that no longer allows it to be outstanding — wait_vmcnt<N> lands all but the newest N,
// __syncthreads() all (hipcc drains vmcnt before the barrier), a bare s_barrier none.  Register loads and stores, which also
// count on the hardware, are not queued: they can only make the hardware land MORE DMAs at a wait than the model does, so a
// schedule that is correct here is correct there; one that reads a slot too early reads stale bytes here every time.
// LDS race check (VC_EMU_RACE=1; dynamic LDS only): two waves may touch the same 16 bytes in one barrier epoch only if both
// read.  A read of bytes another wave wrote (or DMA-landed) since the last barrier, a write or a DMA ISSUE over bytes another wave
// read or wrote since the last barrier -> the emulator aborts with the block, the waves and the LDS offset.  Accesses of one wave
// are ordered by its own instruction stream and vmcnt waits (the model above); barriers are __syncthreads() and the bare s_barrier.
// Same-wave write-after-read (VC_EMU_RACE=1): a DMA issued into bytes this lane has read without an lgkmcnt wait (explicit, or the
// one hipcc puts in front of __syncthreads()) in between — the read may still sit in the LDS queue when the DMA's data arrives.
// Conservative: a read whose VALUE was already consumed has completed too, which the emulator cannot see.
extern bool g_race;   // VC_EMU_RACE != 0, read once: the hooks below cost one predictable branch when it is off
void lgkm_wait_slow(int keep_newest);
void lds_read_slow(const void* p);
void lds_write_slow(const void* p, bool dma_issue_only);
inline void lgkm_wait(int keep_newest) { if (g_race) lgkm_wait_slow(keep_newest); }
inline void lds_read(const void* p) { if (g_race) lds_read_slow(p); }
inline void lds_write(const void* p, bool dma_issue_only) { if (g_race) lds_write_slow(p, dma_issue_only); }
void dma_issue(const void* src, void* dst, const void* wave_base);
void dma_wait(int keep_newest);
void block_barrier();
void wave_sync();
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
}  // namespace vc_emu


// ---- minimal HIP runtime shims so that the host engine (engine.hip) also runs under the emulator ----------
#include <chrono>
// the device's constant-rate wall clock (vc_device.h vc_wall_clock): 100 MHz ticks of the host's steady clock
inline unsigned long long vc_emu_wall_clock() {
    return (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10;
}
typedef int hipError_t;
static const hipError_t hipSuccess = 0;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal };
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
// Guard pages (default; VC_EMU_GUARD=0 switches to plain padded allocations): every device allocation ends (to 16 bytes) at an
// inaccessible page, so a kernel that reads or writes past its buffer faults instead of silently touching the allocator's slack —
// what a 256-byte padding, or the page granularity of a real hipMalloc, would hide.
struct EmuGuardTable {
    std::mutex mu;
    std::map<void*, std::pair<void*, size_t>> live;   // user pointer -> (mapping base, mapping bytes)
};
inline EmuGuardTable& emu_guard_table() {
    static EmuGuardTable t;
    return t;
}
inline bool emu_guard_on() {
    static const bool on = !(getenv("VC_EMU_GUARD") && atoi(getenv("VC_EMU_GUARD")) == 0);
    return on;
}
inline void* emu_alloc(size_t n) {
    if (!emu_guard_on()) return aligned_alloc(256, (n + 255) / 256 * 256 + 256);
    // VC_EMU_GUARD=2: the mirror image — the buffer STARTS right behind an inaccessible page (underruns fault)
    static const bool front = getenv("VC_EMU_GUARD") && atoi(getenv("VC_EMU_GUARD")) == 2;
    const size_t page = 4096, need = ((n ? n : 1) + 15) / 16 * 16, total = (need + page - 1) / page * page + page;
    char* base = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (base == (char*)MAP_FAILED) return nullptr;
    mprotect(front ? base : base + total - page, page, PROT_NONE);
    void* user = front ? base + page : base + (total - page) - need;
    // fresh device memory is garbage, not zeros: 0xFF bytes are NaNs in fp32 and bf16 and -1 as an index, so a kernel that reads
    // what nobody wrote poisons its output instead of passing on the mapping's zero fill (VC_EMU_POISON=0: leave the zeros)
    static const bool poison = !(getenv("VC_EMU_POISON") && atoi(getenv("VC_EMU_POISON")) == 0);
    if (poison) memset(user, 0xFF, need);
    EmuGuardTable& t = emu_guard_table();
    std::lock_guard<std::mutex> lk(t.mu);
    t.live[user] = {base, total};
    return user;
}
inline void emu_free(void* p) {
    if (!p) return;
    if (!emu_guard_on()) return free(p);
    EmuGuardTable& t = emu_guard_table();
    std::lock_guard<std::mutex> lk(t.mu);
    auto it = t.live.find(p);
    if (it == t.live.end()) return;
    munmap(it->second.first, it->second.second);
    t.live.erase(it);
}
inline hipError_t hipMalloc(void** p, size_t n) { *p = emu_alloc(n); return *p ? 0 : 1; }
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { emu_free(p); return 0; }
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
static const unsigned hipStreamNonBlocking = 1;
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (void*)0x1; return 0; }
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
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
inline hipError_t hipEventQuery(hipEvent_t) { return 0; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = aligned_alloc(256, (n + 255) / 256 * 256 + 256); return *p ? 0 : 1; }
inline hipError_t hipHostFree(void* p) { free(p); return 0; }
inline hipError_t hipMemset2DAsync(void* d, size_t pitch, int v, size_t w, size_t h, hipStream_t) {
    for (size_t r = 0; r < h; ++r) memset((char*)d + r * pitch, v, w);
    return 0;
}
namespace vc_emu {
struct Graph { std::vector<std::function<void()>> ops; };
extern thread_local Graph* g_capturing;  // per host thread, like hipStreamCaptureModeThreadLocal
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

#define threadIdx (vc_emu::g_cur->tidx)
#define blockIdx (vc_emu::g_cur->blk->bidx)
#define blockDim (vc_emu::g_cur->blk->bdim)
#define gridDim (vc_emu::g_cur->blk->gdim)

inline void __syncthreads() {
    vc_emu::dma_wait(0);
    vc_emu::lgkm_wait(0);
    vc_emu::block_barrier();
}
inline float __expf(float x) { return expf(x); }
inline float __logf(float x) { return logf(x); }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __fmul_rn(float a, float b) { return a * b; }
inline float __fadd_rn(float a, float b) { return a + b; }

#define VC_DYNAMIC_SMEM(type, name) type* name = reinterpret_cast<type*>(vc_emu::g_cur->blk->dyn_smem)
#define VC_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    vc_emu::launch(grid, block, shmem, [=]() { kernel(__VA_ARGS__); })

namespace vc {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

inline int lane_id() { return vc_emu::g_cur->lane; }

template <class T> inline T shfl_xor(T v, int mask) {
    static_assert(sizeof(T) <= 64, "xchg slot");
    auto& w = vc_emu::g_cur->blk->waves[vc_emu::g_cur->wave];
    memcpy(w.xchg[vc_emu::g_cur->lane], &v, sizeof(T));
    vc_emu::wave_sync();
    T r;
    memcpy(&r, w.xchg[(vc_emu::g_cur->lane ^ mask) & 63], sizeof(T));
    vc_emu::wave_sync();
    return r;
}
template <class T> inline T shfl(T v, int src) {
    auto& w = vc_emu::g_cur->blk->waves[vc_emu::g_cur->wave];
    memcpy(w.xchg[vc_emu::g_cur->lane], &v, sizeof(T));
    vc_emu::wave_sync();
    T r;
    memcpy(&r, w.xchg[src & 63], sizeof(T));
    vc_emu::wave_sync();
    return r;
}

// v_mfma_f32_16x16x4_f32: A[i][k] in lane i+16k, B[k][j] in lane j+16k, D[i][j] in lane j+16*(i/4) reg i%4
inline f32x4 mfma16_f32(float a, float b, f32x4 c) {
    auto& w = vc_emu::g_cur->blk->waves[vc_emu::g_cur->wave];
    const int lane = vc_emu::g_cur->lane;
    memcpy(w.xchg[lane], &a, 4);
    memcpy(w.xchg[lane] + 4, &b, 4);
    vc_emu::wave_sync();
    const int j = lane & 15;
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int i = (lane >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            float af, bf;
            memcpy(&af, w.xchg[i + 16 * k], 4);
            memcpy(&bf, w.xchg[j + 16 * k] + 4, 4);
            acc = fmaf(af, bf, acc);
        }
        d[r] = acc;
    }
    vc_emu::wave_sync();
    return d;
}

// v_mfma_scale_f32_16x16x128_f8f6f4, e4m3 x e4m3, unit block scales: A[i][k] in lane i+16*(k/32) byte k%32, B likewise.
inline float emu_e4m3(uint8_t b) {
    const int e = (b >> 3) & 15, m = b & 7;
    const float mag = e == 0 ? (float)m * 0.001953125f : ldexpf(1.f + (float)m * 0.125f, e - 7);
    return (b & 0x80) ? -mag : mag;
}
inline f32x4 mfma16_f8(u32x4 a0, u32x4 a1, u32x4 b0, u32x4 b1, f32x4 c) {
    auto& w = vc_emu::g_cur->blk->waves[vc_emu::g_cur->wave];
    const int lane = vc_emu::g_cur->lane;
    memcpy(w.xchg[lane], &a0, 16);
    memcpy(w.xchg[lane] + 16, &a1, 16);
    memcpy(w.xchg[lane] + 32, &b0, 16);
    memcpy(w.xchg[lane] + 48, &b1, 16);
    vc_emu::wave_sync();
    const int j = lane & 15;
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int i = (lane >> 4) * 4 + r;
        // VC_EMU_F8_BITS=n (experiment): model a datapath that aligns the 128 products to the largest one and keeps n bits
        static const int f8_bits = getenv("VC_EMU_F8_BITS") ? atoi(getenv("VC_EMU_F8_BITS")) : 0;
        float prod[128], pmax = 0.f;
        for (int k = 0; k < 128; ++k) {
            prod[k] = emu_e4m3(w.xchg[i + 16 * (k / 32)][k % 32]) * emu_e4m3(w.xchg[j + 16 * (k / 32)][32 + k % 32]);
            pmax = fmaxf(pmax, fabsf(prod[k]));
        }
        float acc = 0.f;
        if (f8_bits > 0 && pmax > 0.f) {
            int ex;
            frexpf(pmax, &ex);
            const float q = ldexpf(1.f, ex - f8_bits);
            double a2 = 0.0;
            for (int k = 0; k < 128; ++k) a2 += (double)(truncf(prod[k] / q) * q);
            acc = (float)a2;
        } else {
            for (int k = 0; k < 128; ++k) acc += prod[k];
        }
        d[r] = c[r] + acc;
    }
    vc_emu::wave_sync();
    return d;
}

// one 16-bit MFMA operand element as a float: bf16, or fp16 in the -DVC_F16 build (vc_device.h)
inline float emu_operand(uint16_t b) {
#ifdef VC_F16
    const uint32_t sign = (uint32_t)(b & 0x8000u) << 16, e = (b >> 10) & 31u, m = b & 0x3FFu;
    uint32_t u;
    float f;
    if (e == 0) {
        f = (float)m * 5.9604644775390625e-08f;
        memcpy(&u, &f, 4);
        u |= sign;
    } else if (e == 31) u = sign | 0x7F800000u | (m << 13);
    else u = sign | ((e + 112u) << 23) | (m << 13);
    memcpy(&f, &u, 4);
    return f;
#else
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
#endif
}
// v_mfma_f32_32x32x16_bf16: A[i][k] in lane i + 32 (k / 8) elem k % 8, B likewise by column, D[i][j] in lane j + 32 ((i / 4) % 2),
// reg i % 4 + 4 (i / 8)
inline f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
    auto& w = vc_emu::g_cur->blk->waves[vc_emu::g_cur->wave];
    const int lane = vc_emu::g_cur->lane;
    memcpy(w.xchg[lane], &a, 16);
    memcpy(w.xchg[lane] + 16, &b, 16);
    vc_emu::wave_sync();
    const int j = lane & 31;
    f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float acc = 0.f;
        for (int k = 0; k < 16; ++k) {
            uint16_t ab, bb;
            memcpy(&ab, w.xchg[i + 32 * (k / 8)] + 2 * (k % 8), 2);
            memcpy(&bb, w.xchg[j + 32 * (k / 8)] + 16 + 2 * (k % 8), 2);
            acc += emu_operand(ab) * emu_operand(bb);
        }
        d[r] = c[r] + acc;
    }
    vc_emu::wave_sync();
    return d;
}

// v_mfma_f32_16x16x32_bf16 under the assumed fragment maps (see vc_device.h header).
inline f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
    auto& w = vc_emu::g_cur->blk->waves[vc_emu::g_cur->wave];
    const int lane = vc_emu::g_cur->lane;
    memcpy(w.xchg[lane], &a, 16);
    memcpy(w.xchg[lane] + 16, &b, 16);
    vc_emu::wave_sync();
    const int j = lane & 15;
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int i = (lane >> 4) * 4 + r;
        float acc = 0.f;
        for (int k = 0; k < 32; ++k) {
            uint16_t ab, bb;
            memcpy(&ab, w.xchg[i + 16 * (k / 8)] + 2 * (k % 8), 2);
            memcpy(&bb, w.xchg[j + 16 * (k / 8)] + 16 + 2 * (k % 8), 2);
            acc += emu_operand(ab) * emu_operand(bb);
        }
        d[r] = c[r] + acc;
    }
    vc_emu::wave_sync();
    return d;
}
}  // namespace vc
