// engine_ctx.h — the context object behind `vc_ctx*` (include/vcoder_hip.h), shared by engine.hip and comm.hip.
#pragma once
#include <string>

#include "kernels.h"

struct vc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
};
