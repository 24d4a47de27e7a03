// corun.hip — does a lean (<= 48 VGPR, <= 24 KiB LDS) streaming kernel get co-scheduled on CUs that are running the
// 256x256 8-phase GEMM (464 of 512 VGPRs per SIMD, 128 of 160 KiB LDS), and what HBM rate does it reach there?
// build: hipcc --offload-arch=gfx950 -O3 tools/experiments/corun.hip -o tools/experiments/corun -Lvcoder_amd/lib -lvcoder_hip
// run  : LD_LIBRARY_PATH=vcoder_amd/lib tools/experiments/corun
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

extern "C" void vck_gemm(const uint16_t* A, const uint16_t* W, const float* bias, void* out, int M, int N, int K, int lda,
                         int ldw, int ldo, int epi, void* stream);

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// R-slot wave-private LDS-DMA ring: each wave streams its share of `src` (1 KiB per instruction) and xors it up
template <int R, int AUX>
__global__ __launch_bounds__(256) void lean_stream(const char* __restrict__ src, size_t kib_per_wave, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* my = ring + wave * R * 1024;
    const size_t gw = (size_t)blockIdx.x * 4 + wave;
    const char* p = src + gw * kib_per_wave * 1024 + lane * 16;
    u32x4 acc = {0, 0, 0, 0};
    const int n = (int)kib_per_wave;
#pragma unroll
    for (int i = 0; i < R; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (size_t)(i < n ? i : n - 1) * 1024),
                                         (__attribute__((address_space(3))) void*)(my + i * 1024), 16, 0, AUX);
    for (int i = 0; i < n; i += R) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R - 1) : "memory");
            const u32x4 v = *reinterpret_cast<const u32x4*>(my + r * 1024 + lane * 16);
            acc ^= v;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int nx = i + r + R;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (size_t)(nx < n ? nx : n - 1) * 1024),
                                             (__attribute__((address_space(3))) void*)(my + r * 1024), 16, 0, AUX);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}

template <int R, int AUX>
int run(uint16_t* A, uint16_t* W, void* out, char* src, unsigned* flag, size_t bytes, hipStream_t sg, hipStream_t ss, hipEvent_t e0,
        hipEvent_t e1, hipEvent_t g0, hipEvent_t g1, int M, int N, int K);

int main() {
    const int M = 9728, N = 22016, K = 4096;
    uint16_t *A, *W;
    void* out;
    CHECK(hipMalloc(&A, (size_t)M * K * 2));
    CHECK(hipMalloc(&W, (size_t)N * K * 2));
    CHECK(hipMalloc(&out, (size_t)M * N * 2));
    CHECK(hipMemset(A, 0x11, (size_t)M * K * 2));
    CHECK(hipMemset(W, 0x22, (size_t)N * K * 2));
    const size_t bytes = (size_t)2 << 30;  // 2 GiB streamed per launch
    char* src;
    unsigned* flag;
    CHECK(hipMalloc(&src, bytes));
    CHECK(hipMalloc(&flag, 4));
    CHECK(hipMemset(src, 1, bytes));
    hipStream_t sg, ss;
    CHECK(hipStreamCreateWithFlags(&sg, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&ss, hipStreamNonBlocking));
    hipEvent_t e0, e1, g0, g1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&g0)); CHECK(hipEventCreate(&g1));
    return run<6, 0>(A, W, out, src, flag, bytes, sg, ss, e0, e1, g0, g1, M, N, K) | run<6, 2>(A, W, out, src, flag, bytes, sg, ss, e0, e1, g0, g1, M, N, K) |
           run<3, 2>(A, W, out, src, flag, bytes, sg, ss, e0, e1, g0, g1, M, N, K);
}

template <int R, int AUX>
int run(uint16_t* A, uint16_t* W, void* out, char* src, unsigned* flag, size_t bytes, hipStream_t sg, hipStream_t ss, hipEvent_t e0,
        hipEvent_t e1, hipEvent_t g0, hipEvent_t g1, int M, int N, int K) {
    printf("---- R=%d aux=%d\n", R, AUX);
    const int lds = 4 * R * 1024;
    const int wgs = 256 * 8;                       // 8 workgroups per CU worth of work
    const size_t kib_per_wave = bytes / 1024 / ((size_t)wgs * 4);
    auto stream_once = [&]() { hipLaunchKernelGGL((lean_stream<R, AUX>), dim3(wgs), dim3(256), lds, ss, src, kib_per_wave, flag); };
    auto gemm_once = [&]() { vck_gemm(A, W, nullptr, out, M, N, K, K, K, N / 2, 5, sg); };
    float ms;
    // --- short launches: the qkv GEMV's geometry (768 WGs x 4 waves x 32 KiB = 96 MiB per launch), rotating through src
    for (int cfg = 0; cfg < 3; ++cfg) {
        const int swg = cfg == 0 ? 768 : cfg == 1 ? 1376 : 256;
        const size_t skib = cfg == 0 ? 32 : cfg == 1 ? 32 : 128;
        const size_t per = (size_t)swg * 4 * skib * 1024;
        auto short_once = [&](int it) {
            hipLaunchKernelGGL((lean_stream<R, AUX>), dim3(swg), dim3(256), lds, ss, src + (size_t)(it % 8) * per, skib, flag);
        };
        for (int i = 0; i < 8; ++i) short_once(i);
        CHECK(hipStreamSynchronize(ss));
        CHECK(hipEventRecord(e0, ss)); for (int i = 0; i < 40; ++i) short_once(i); CHECK(hipEventRecord(e1, ss));
        CHECK(hipStreamSynchronize(ss)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("short launch %4d WGs x 4 waves x %3zu KiB = %.1f MB: %.1f us  %.2f TB/s\n", swg, skib, per / 1e6, ms / 40 * 1e3,
               per / (ms / 40) / 1e9);
    }
    // --- alone
    stream_once(); CHECK(hipStreamSynchronize(ss));
    CHECK(hipEventRecord(e0, ss)); for (int i = 0; i < 4; ++i) stream_once(); CHECK(hipEventRecord(e1, ss));
    CHECK(hipStreamSynchronize(ss)); CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("lean stream alone : %.2f TB/s (%d B LDS/WG, R=%d)\n", 4.0 * bytes / ms / 1e9, lds, R);
    gemm_once(); CHECK(hipStreamSynchronize(sg));
    CHECK(hipEventRecord(g0, sg)); for (int i = 0; i < 8; ++i) gemm_once(); CHECK(hipEventRecord(g1, sg));
    CHECK(hipStreamSynchronize(sg)); CHECK(hipEventElapsedTime(&ms, g0, g1));
    const double gemm_alone = ms / 8;
    printf("gemm alone        : %.3f ms  %.0f TFLOP/s\n", gemm_alone, 2.0 * M * N * K / gemm_alone / 1e9);
    // --- together: GEMMs run continuously while the stream kernel is timed
    CHECK(hipEventRecord(g0, sg));
    for (int i = 0; i < 12; ++i) gemm_once();
    CHECK(hipEventRecord(g1, sg));
    CHECK(hipEventRecord(e0, ss)); for (int i = 0; i < 4; ++i) stream_once(); CHECK(hipEventRecord(e1, ss));
    CHECK(hipStreamSynchronize(ss));
    const bool gemm_still_running = hipEventQuery(g1) == hipErrorNotReady;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("lean stream co-run: %.2f TB/s (gemm still running at the end: %s)\n", 4.0 * bytes / ms / 1e9,
           gemm_still_running ? "yes" : "NO - lengthen");
    const double stream_ms = ms;
    CHECK(hipStreamSynchronize(sg)); CHECK(hipEventElapsedTime(&ms, g0, g1));
    printf("12 gemms with the stream beside them: %.3f ms each on average (alone %.3f); stream ran %.1f ms of %.1f\n", ms / 12,
           gemm_alone, stream_ms, ms);
    return 0;
}
