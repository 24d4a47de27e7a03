// TEST-ONLY: runtime of the fiber-per-lane emulator (see hip_emu.h).
//
// A launch spawns up to NWORKERS OS threads; each takes workgroups round-robin.  The threads of one workgroup are
// fibers with a hand-rolled x86-64 context switch (callee-saved registers + stack pointer), scheduled
// round-robin; a fiber yields when it waits at __syncthreads() or at a wave-level collective (shuffle / MFMA).
// Like the hardware, barriers count only threads that have not exited.
#include "hip_emu.h"

#include <stdio.h>
#include <sys/mman.h>

extern "C" void vc_emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl vc_emu_switch
.type vc_emu_switch,@function
vc_emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
)");

namespace vc_emu {
thread_local Fiber* g_cur = nullptr;
thread_local Graph* g_capturing = nullptr;

static const size_t STACK_BYTES = 256 * 1024;

void yield_to_scheduler() {
    Fiber* f = g_cur;
    vc_emu_switch(&f->sp, f->blk->sched_sp);
}

void block_barrier() {
    BlockCtx* b = g_cur->blk;
    const unsigned gen = b->bar_gen;
    if (++b->bar_arrived >= b->alive) {
        b->bar_arrived = 0;
        ++b->bar_gen;
        ++b->progress;
        return;
    }
    while (b->bar_gen == gen) yield_to_scheduler();
}

void wave_sync() {
    BlockCtx* b = g_cur->blk;
    WaveCtx& w = b->waves[g_cur->wave];
    const unsigned gen = w.gen;
    if (++w.arrived >= b->wave_alive[g_cur->wave]) {
        w.arrived = 0;
        ++w.gen;
        ++b->progress;
        return;
    }
    while (w.gen == gen) yield_to_scheduler();
}

bool g_race = getenv("VC_EMU_RACE") && atoi(getenv("VC_EMU_RACE")) != 0;
static bool race_on() { return g_race; }
static LdsShadow* shadow_of(Fiber* f, const void* p) {
    BlockCtx* b = f->blk;
    if (!b->shadow) return nullptr;
    const size_t off = (size_t)((const char*)p - b->dyn_smem);
    if ((const char*)p < b->dyn_smem || off >= b->dyn_bytes) return nullptr;
    return b->shadow + off / 16;
}
static void race_abort(Fiber* f, const void* p, const char* what, int other) {
    BlockCtx* b = f->blk;
    fprintf(stderr, "emu: LDS race in block (%u,%u,%u): %s — wave %d lane %d vs wave %d, dynamic LDS offset %zu, barrier epoch %u\n",
            b->bidx.x, b->bidx.y, b->bidx.z, what, f->wave, f->lane, other, (size_t)((const char*)p - b->dyn_smem), b->bar_gen);
    abort();
}
void lgkm_wait_slow(int keep_newest) {
    Fiber* f = g_cur;
    if (!f) return;
    if ((int)(f->lds_rd_tail - f->lds_rd_head) > keep_newest) f->lds_rd_head = f->lds_rd_tail - (unsigned)keep_newest;
}
void lds_read_slow(const void* p) {
    Fiber* f = g_cur;
    if (!f || !f->blk->shadow) return;
    LdsShadow* s = shadow_of(f, p);
    if (!s) return;
    if (f->lds_rd_tail - f->lds_rd_head >= 64) ++f->lds_rd_head;
    f->lds_rd[f->lds_rd_tail++ % 64] = (const void*)((uintptr_t)p & ~(uintptr_t)15);
    const unsigned e = f->blk->bar_gen + 1;   // epochs count from 1: 0 = never
    if (s->w_epoch == e && s->w_wave >= 0 && s->w_wave != f->wave) race_abort(f, p, "read of bytes another wave wrote since the last barrier", s->w_wave);
    if (s->r_epoch != e) {
        s->r_epoch = e;
        s->r_wave = (short)f->wave;
    } else if (s->r_wave != f->wave) {
        s->r_wave = -2;
    }
}
void lds_write_slow(const void* p, bool dma_issue_only) {
    Fiber* f = g_cur;
    if (!f || !f->blk->shadow) return;
    LdsShadow* s = shadow_of(f, p);
    if (!s) return;
    const unsigned e = f->blk->bar_gen + 1;
    if (s->r_epoch == e && s->r_wave != -1 && s->r_wave != f->wave)
        race_abort(f, p, dma_issue_only ? "DMA issued over bytes another wave read since the last barrier" : "write over bytes another wave read since the last barrier", s->r_wave);
    if (s->w_epoch == e && s->w_wave >= 0 && s->w_wave != f->wave)
        race_abort(f, p, dma_issue_only ? "DMA issued over bytes another wave wrote since the last barrier" : "write over bytes another wave wrote since the last barrier", s->w_wave);
    if (!dma_issue_only) {
        s->w_epoch = e;
        s->w_wave = (short)f->wave;
    }
}

static bool dma_deferred() {
    static const bool on = !(getenv("VC_EMU_DMA") && atoi(getenv("VC_EMU_DMA")) == 0);
    return on;
}
void dma_issue(const void* src, void* dst, const void* wave_base) {
    Fiber* f = g_cur;
    if (f) {   // the LDS base comes from M0 on the hardware: it has to be wave-uniform
        WaveCtx& w = f->blk->waves[f->wave];
        const unsigned seq = ++f->dma_seq, slot = seq % 256;
        if (w.dma_base_seq[slot] != seq) {
            w.dma_base_seq[slot] = seq;
            w.dma_base[slot] = wave_base;
        } else if (w.dma_base[slot] != wave_base) {
            fprintf(stderr, "emu: LDS-DMA %u of wave %d in block (%u,%u,%u): lane %d passes LDS base %p, an earlier lane %p — the base "
                            "is taken from M0 and must be wave-uniform\n", seq, f->wave, f->blk->bidx.x, f->blk->bidx.y, f->blk->bidx.z,
                    f->lane, wave_base, w.dma_base[slot]);
            abort();
        }
    }
    lds_write(dst, true);
    if (f && f->blk->shadow)
        for (unsigned i = f->lds_rd_head; i != f->lds_rd_tail; ++i)
            if (f->lds_rd[i % 64] == (const void*)((uintptr_t)dst & ~(uintptr_t)15)) {
                fprintf(stderr, "emu: block (%u,%u,%u) wave %d lane %d: DMA issued into dynamic LDS offset %zu that this lane read without an "
                                "lgkmcnt wait in between\n", f->blk->bidx.x, f->blk->bidx.y, f->blk->bidx.z, f->wave, f->lane,
                        (size_t)((const char*)dst - f->blk->dyn_smem));
                abort();
            }
    if (!f || !dma_deferred()) {
        memcpy(dst, src, 16);
        lds_write(dst, false);
        return;
    }
    if (f->dma_tail - f->dma_head >= 128) {   // cannot happen with a sane schedule: land the oldest
        const PendingDma& d = f->dma[f->dma_head++ % 128];
        memcpy(d.dst, d.src, 16);
        lds_write(d.dst, false);
    }
    f->dma[f->dma_tail++ % 128] = PendingDma{src, dst};
}
void dma_wait(int keep_newest) {
    Fiber* f = g_cur;
    if (!f) return;
    while ((int)(f->dma_tail - f->dma_head) > keep_newest) {
        const PendingDma& d = f->dma[f->dma_head++ % 128];
        memcpy(d.dst, d.src, 16);
        lds_write(d.dst, false);
    }
}

static void fiber_finished(Fiber* f) {
    BlockCtx* b = f->blk;
    f->done = true;
    --b->alive;
    --b->wave_alive[f->wave];
    ++b->progress;
    // a thread that exits releases barriers the remaining threads are all waiting at
    if (b->alive > 0 && b->bar_arrived >= b->alive && b->bar_arrived > 0) {
        b->bar_arrived = 0;
        ++b->bar_gen;
    }
    WaveCtx& w = b->waves[f->wave];
    if (b->wave_alive[f->wave] > 0 && w.arrived >= b->wave_alive[f->wave] && w.arrived > 0) {
        w.arrived = 0;
        ++w.gen;
    }
}

static void fiber_entry() {
    Fiber* f = g_cur;
    (*f->blk->body)();
    fiber_finished(f);
    void* dummy;
    vc_emu_switch(&dummy, f->blk->sched_sp);
    abort();  // never resumed
}

static void run_block(BlockCtx* b, std::vector<Fiber>& fibers) {
    const int n = b->nthreads;
    b->alive = n;
    b->bar_arrived = 0;
    if (b->shadow)
        for (size_t i = 0; i <= b->dyn_bytes / 16; ++i) b->shadow[i] = LdsShadow{0u, 0u, (short)-1, (short)-1};
    for (int w = 0; w < n / 64; ++w) {
        b->wave_alive[w] = 64;
        b->waves[w].arrived = 0;
        memset(b->waves[w].dma_base_seq, 0, sizeof(b->waves[w].dma_base_seq));
    }
    for (int t = 0; t < n; ++t) {
        Fiber& f = fibers[t];
        f.done = false;
        f.dma_head = f.dma_tail = 0;
        f.dma_seq = 0;
        f.lds_rd_head = f.lds_rd_tail = 0;
        f.blk = b;
        f.lane = t & 63;
        f.wave = t >> 6;
        f.tidx = dim3(t % b->bdim.x, (t / b->bdim.x) % b->bdim.y, t / (b->bdim.x * b->bdim.y));
        // initial frame: 6 callee-saved slots + return address = fiber_entry; rsp % 16 == 8 at function entry
        uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15;
        void** sp = (void**)(top - 8);
        *--sp = (void*)&fiber_entry;
        for (int i = 0; i < 6; ++i) *--sp = nullptr;
        f.sp = sp;
    }
    // VC_EMU_ORDER: the order in which the scheduler visits the waves of a block in each pass — 0 ascending (default), 1 descending,
    // >= 2 a pseudo-random permutation per pass (seed).  Fibers yield only at barriers and wave collectives, so the wave order IS
    // the interleaving: a kernel that relies on a barrier it does not have computes differently under another order.
    static const int order_mode = getenv("VC_EMU_ORDER") ? atoi(getenv("VC_EMU_ORDER")) : 0;
    const int nw = n / 64;
    int worder[16];
    for (int w = 0; w < nw; ++w) worder[w] = w;
    unsigned long lcg = 0x9E3779B97F4A7C15ul * (unsigned long)(order_mode + 1) + b->bidx.x * 131ul + b->bidx.y * 7919ul + b->bidx.z;
    while (b->alive > 0) {
        const unsigned long before = b->progress;
        if (order_mode == 1) {
            for (int w = 0; w < nw; ++w) worder[w] = nw - 1 - w;
        } else if (order_mode >= 2) {
            for (int w = nw - 1; w > 0; --w) {   // Fisher-Yates
                lcg = lcg * 6364136223846793005ul + 1442695040888963407ul;
                const int j = (int)((lcg >> 33) % (unsigned long)(w + 1));
                const int tmp = worder[w];
                worder[w] = worder[j];
                worder[j] = tmp;
            }
        }
        for (int wi = 0; wi < nw; ++wi)
            for (int l = 0; l < 64; ++l) {
                Fiber& f = fibers[worder[wi] * 64 + l];
                if (f.done) continue;
                g_cur = &f;
                vc_emu_switch(&b->sched_sp, f.sp);
            }
        if (b->progress == before && b->alive > 0) {
            fprintf(stderr, "emu: deadlock in block (%u,%u,%u): %d threads alive, barrier %d arrived\n", b->bidx.x,
                    b->bidx.y, b->bidx.z, b->alive, b->bar_arrived);
            abort();
        }
    }
    g_cur = nullptr;
}

static int n_workers() {
    static int n = [] {
        const char* e = getenv("VC_EMU_WORKERS");
        int v = e ? atoi(e) : (int)std::thread::hardware_concurrency();
        return std::max(1, std::min(v, 32));
    }();
    return n;
}

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
    const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    const int workers = (int)std::min<size_t>(n_workers(), nblocks);
    auto work = [&](int wid) {
        char* stacks = (char*)mmap(nullptr, STACK_BYTES * nthreads, PROT_READ | PROT_WRITE,
                                   MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (stacks == MAP_FAILED) abort();
        std::vector<Fiber> fibers(nthreads);
        for (int t = 0; t < nthreads; ++t) fibers[t].stack = stacks + (size_t)t * STACK_BYTES;
        BlockCtx* b = new BlockCtx();
        b->nthreads = nthreads;
        b->bdim = block;
        b->gdim = grid;
        b->body = &body;
        b->dyn_smem = (char*)emu_alloc(shmem);   // dynamic LDS of the launch: guard page behind it like every device buffer
        b->dyn_bytes = shmem;
        b->shadow = (race_on() && shmem > 0) ? (LdsShadow*)malloc((shmem / 16 + 1) * sizeof(LdsShadow)) : nullptr;
        for (size_t i = wid; i < nblocks; i += workers) {
            b->bidx = dim3((unsigned)(i % grid.x), (unsigned)((i / grid.x) % grid.y), (unsigned)(i / ((size_t)grid.x * grid.y)));
            run_block(b, fibers);
        }
        emu_free(b->dyn_smem);
        free(b->shadow);
        delete b;
        munmap(stacks, STACK_BYTES * nthreads);
    };
    if (workers == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int w = 0; w < workers; ++w) th.emplace_back(work, w);
        for (auto& t : th) t.join();
    }
}
}  // namespace vc_emu
