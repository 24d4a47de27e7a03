// TEST-ONLY: runtime of the thread-per-lane emulator (see hip_emu.h).
#include "hip_emu.h"
#include <stdio.h>

thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
namespace vc_emu {
thread_local BlockCtx* g_ctx = nullptr;
thread_local int g_lane = 0, g_wave = 0;
Graph* g_capturing = nullptr;

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    if (g_capturing) {  // stream capture: record, run at hipGraphLaunch
        Graph* g = g_capturing;
        g->ops.push_back([=]() { launch(grid, block, shmem, body); });
        return;
    }
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads % 64 != 0 || nthreads > 1024) {
        fprintf(stderr, "emu: block size %d must be a multiple of 64 (<=1024)\n", nthreads);
        abort();
    }
    BlockCtx* ctx = new BlockCtx();
    ctx->nthreads = nthreads;
    pthread_barrier_init(&ctx->bar, nullptr, nthreads);
    const int nwaves = nthreads / 64;
    for (int w = 0; w < nwaves; ++w) pthread_barrier_init(&ctx->waves[w].bar, nullptr, 64);
    ctx->dyn_smem = (char*)aligned_alloc(256, (shmem + 255) / 256 * 256 + 256);
    std::vector<std::thread> th;
    th.reserve(nthreads);
    for (int t = 0; t < nthreads; ++t) {
        th.emplace_back([=]() {
            g_ctx = ctx;
            g_lane = t & 63;
            g_wave = t >> 6;
            ::blockDim = block;
            ::gridDim = grid;
            ::threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            for (unsigned bz = 0; bz < grid.z; ++bz)
                for (unsigned by = 0; by < grid.y; ++by)
                    for (unsigned bx = 0; bx < grid.x; ++bx) {
                        ::blockIdx = dim3(bx, by, bz);
                        body();
                        pthread_barrier_wait(&ctx->bar);  // blocks run one after another
                    }
        });
    }
    for (auto& t : th) t.join();
    pthread_barrier_destroy(&ctx->bar);
    for (int w = 0; w < nwaves; ++w) pthread_barrier_destroy(&ctx->waves[w].bar);
    free(ctx->dyn_smem);
    delete ctx;
}
}  // namespace vc_emu
