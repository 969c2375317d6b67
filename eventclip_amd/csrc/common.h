// Shared host-side helpers for libeventclip_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/eventclip_hip.h"

namespace ec {

// thread-local message behind ec_last_error()
char *err_buf();
int fail(int code, const char *fmt, ...);

#define EC_CHECK_HIP(expr)                                                                  \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            return ::ec::fail(EC_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                   \
                              hipGetErrorString(_e), __FILE__, __LINE__);                   \
    } while (0)

#define EC_REQUIRE(cond, ...)                                                               \
    do {                                                                                    \
        if (!(cond)) return ::ec::fail(EC_ERR_INVALID, __VA_ARGS__);                        \
    } while (0)

// Optional launch profiler behind ec_profile_begin / ec_profile_end: when enabled, every
// instrumented launch is bracketed by a pair of hipEvents on its own stream.  Classes
// are the kernel symbols a rocprofv3 --kernel-trace lists.
enum ProfClass {
    PROF_EVENTS = 0, PROF_PREPROCESS, PROF_PATCHIFY, PROF_GEMM_STORE16, PROF_GEMM_GELU16,
    PROF_GEMM_RESID32, PROF_GEMM_STORE32, PROF_LAYERNORM, PROF_ATTENTION, PROF_EMBED, PROF_CLASSIFY,
    PROF_ADAPTER, PROF_GEMM_DW, PROF_ATTENTION_BWD, PROF_LN_BWD, PROF_TRANSPOSE, PROF_REDUCE, PROF_SGEMM,
    PROF_OPTIMIZER, PROF_PACK, PROF_NCLASS
};
struct ProfScope {
    ProfScope(int cls, hipStream_t s, double flops, double bytes);
    ~ProfScope();
    int slot;
    hipStream_t stream;
};

// hipFuncAttributeMaxDynamicSharedMemorySize for `kern` on the CURRENT device, set once per
// (kernel, device) and safe to call from several threads; returns EC_OK or EC_ERR_HIP.
int ensure_dynamic_lds(const void *kern, int bytes);
// compute units of the current device (cached per device); 0 on error
int cu_count();

// ec_fs_text_loss_grad with the image features optionally compact over the valid views (train.hip; shared with
// the fine-tuning head in vit_train.hip)
int fs_text_loss_grad(const float *img_feats, const int32_t *row_idx, const uint8_t *valid, const int32_t *labels,
                      const float *text_param, int B, int T, int D, int K, float logit_scale, int agg,
                      int use_probs_loss, float *loss, float *grad_text, float *agg_logits, void *workspace,
                      size_t workspace_bytes, ec_stream_t stream);

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline long ceil_div(long a, long b) { return (a + b - 1) / b; }

}  // namespace ec
