// comm.hip — the one exchange of the data-parallel hot path: an all-gather of the generated token ids across the GPUs of a
// node (counterpart of the reference's per-GPU answer files + `cat`, scripts/v1_5/eval/cost_depth.sh:10-34).
//
// RCCL over xGMI, called directly through its C API (ncclAllGather on the context's stream).  librccl is bound at run time
// with dlopen, so that libvcoder_hip.so itself links against nothing but the HIP runtime and a single-GPU process never
// loads it.  The message is 4 * n bytes per rank (512 ids = 2 KiB for a batch of 8 x 64): latency-bound, the xGMI link
// bandwidth is irrelevant, so there is nothing to bucket or overlap.
#include <dlfcn.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>

#include "../../include/vcoder_hip.h"
#include "engine_ctx.h"

#define VC_API extern "C" __attribute__((visibility("default")))

namespace {

struct Uid {
    char internal[128];  // NCCL_UNIQUE_ID_BYTES
};

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, /* ncclUniqueId by value */ Uid, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;
};
constexpr int kNcclInt32 = 2;  // ncclDataType_t::ncclInt32

void rccl_bind(Rccl& r) {
#ifdef VC_EMU
    r.why = "the CPU emulator build has no RCCL";
#else
    std::string errs;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.handle) break;
        const char* e = dlerror();  // one call: it returns the message AND clears it
        errs += std::string(errs.empty() ? "" : "; ") + (e ? e : "?");
    }
    if (!r.handle) {
        r.why = "librccl.so could not be loaded: " + errs;
        return;
    }
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.handle, "ncclAllGather"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy) {
        r.why = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy";
        r.handle = nullptr;
    }
#endif
}

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] { rccl_bind(r); });
    return r;
}

}  // namespace

struct vc_comm {
    vc_ctx* ctx = nullptr;
    int rank = 0, world = 1;
    void* nccl = nullptr;  // ncclComm_t
    int32_t *send = nullptr, *recv = nullptr;
    size_t cap = 0;  // ids per rank the device buffers hold
};

static int fail(vc_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}

/* 1 when this communicator's gathers run through RCCL (world > 1, or a single rank created under VC_COMM_FORCE_RCCL=1) */
VC_API int vc_comm_uses_rccl(vc_comm* c) { return c && c->nccl ? 1 : 0; }

/* rank 0 creates the 128-byte RCCL unique id and hands it to the other ranks through any side channel (bench.py: the
 * torch.distributed store); world 1 needs none */
VC_API int vc_comm_unique_id(vc_ctx* ctx, void* out128) {
    if (!out128) return VC_ERR_INVALID;
    Rccl& r = rccl();
    if (!r.handle) return fail(ctx, VC_ERR_STATE, r.why);
    Uid id;
    const int rc = r.GetUniqueId(&id);
    if (rc != 0) return fail(ctx, VC_ERR_HIP, std::string("ncclGetUniqueId: ") + (r.GetErrorString ? r.GetErrorString(rc) : "?"));
    memcpy(out128, &id, sizeof id);
    return VC_OK;
}

VC_API int vc_comm_create(vc_ctx* ctx, int rank, int world, const void* unique_id128, vc_comm** out) {
    if (!ctx || !out || world < 1 || rank < 0 || rank >= world) return VC_ERR_INVALID;
    *out = nullptr;
    vc_comm* c = new vc_comm();
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    // VC_COMM_FORCE_RCCL=1: a single rank still binds librccl, creates a communicator of ONE rank and runs its gathers as
    // ncclAllGather on the stream — so that symbol binding, the by-value ncclUniqueId ABI and the stream ordering of the
    // multi-GPU path execute on a one-GPU box (tests/test_gpu_e2e.py::test_token_comm_rccl_forced_world1) before the first
    // 8-GPU run does.  Read per call, not cached: a test sets it for one communicator.
    const char* force_env = getenv("VC_COMM_FORCE_RCCL");
    const bool force = world == 1 && force_env && force_env[0] == '1';
    if (world > 1 || force) {
        Rccl& r = rccl();
        if (!r.handle || (!unique_id128 && !force)) {
            delete c;
            return fail(ctx, VC_ERR_STATE, r.handle ? "a unique id is required for world > 1" : r.why);
        }
        if (hipSetDevice(ctx->device) != hipSuccess) {
            delete c;
            return fail(ctx, VC_ERR_HIP, "hipSetDevice failed");
        }
        Uid id;
        if (unique_id128) {
            memcpy(&id, unique_id128, sizeof id);
        } else {   // forced single rank: nobody to exchange an id with
            const int rc = r.GetUniqueId(&id);
            if (rc != 0) {
                delete c;
                return fail(ctx, VC_ERR_HIP, std::string("ncclGetUniqueId: ") + (r.GetErrorString ? r.GetErrorString(rc) : "?"));
            }
        }
        const int rc = r.CommInitRank(&c->nccl, world, id, rank);
        if (rc != 0) {
            delete c;
            return fail(ctx, VC_ERR_HIP, std::string("ncclCommInitRank: ") + (r.GetErrorString ? r.GetErrorString(rc) : "?"));
        }
    }
    *out = c;
    return VC_OK;
}

/* all-gather of n int32 ids per rank: global[r * n + i] = rank r's local[i].  Host buffers in and out (the ids leave the
 * device once per batch anyway — the caller decodes them to text); enqueued on the context's stream and awaited. */
VC_API int vc_allgather_tokens(vc_comm* c, const int32_t* local, int n, int32_t* global) {
    if (!c || !local || !global || n < 0) return VC_ERR_INVALID;
    if (n == 0) return VC_OK;
    if (c->world == 1 && !c->nccl) {
        memmove(global, local, (size_t)n * 4);
        return VC_OK;
    }
    vc_ctx* ctx = c->ctx;
    Rccl& r = rccl();
    hipStream_t st = ctx->stream;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, VC_ERR_HIP, "hipSetDevice failed");
    if ((size_t)n > c->cap) {
        if (c->send) (void)hipFree(c->send);
        if (c->recv) (void)hipFree(c->recv);
        c->send = c->recv = nullptr;
        c->cap = 0;
        if (hipMalloc(reinterpret_cast<void**>(&c->send), (size_t)n * 4) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&c->recv), (size_t)n * 4 * c->world) != hipSuccess)
            return fail(ctx, VC_ERR_HIP, "hipMalloc of the gather buffers failed");
        c->cap = (size_t)n;
    }
    if (hipMemcpyAsync(c->send, local, (size_t)n * 4, hipMemcpyHostToDevice, st) != hipSuccess)
        return fail(ctx, VC_ERR_HIP, "H2D copy of the local ids failed");
    const int rc = r.AllGather(c->send, c->recv, (size_t)n, kNcclInt32, c->nccl, st);
    if (rc != 0) return fail(ctx, VC_ERR_HIP, std::string("ncclAllGather: ") + (r.GetErrorString ? r.GetErrorString(rc) : "?"));
    if (hipMemcpyAsync(global, c->recv, (size_t)n * 4 * c->world, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return fail(ctx, VC_ERR_HIP, "D2H copy of the gathered ids failed");
    return VC_OK;
}

VC_API void vc_comm_destroy(vc_comm* c) {
    if (!c) return;
    if (c->nccl) (void)rccl().CommDestroy(c->nccl);
    if (c->send) (void)hipFree(c->send);
    if (c->recv) (void)hipFree(c->recv);
    delete c;
}
