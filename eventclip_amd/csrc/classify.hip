// Logits, view aggregation and probabilities (gfx950, fp32).
//
// Replaces the tail of the reference classifiers' forward:
//   ZSCLIPClassifier.forward  models/clip_cls.py:148-154
//     logits = logit_scale * img_feats @ text_feats.T        (:148, image feats NOT normalised)
//     full_logits[valid_masks] = logits                      (:151-152)
//     _aggregate_logits (:104-121): sum | sum / #valid | max with -1e6 on invalid views
//     _aggregate_probs  (:123-129): softmax per view, mask, mean over valid views
//   FSCLIPClassifier.forward  models/clip_cls.py:326-343
//     F.normalize(feats), zero invalid views, the same logits / aggregation.
//
// One 256-thread workgroup per sample.  The sample's <= 16 view features sit in LDS
// (gathered through row_idx, so the ragged `imgs[valid_masks]` batch never needs a
// boolean gather/scatter); thread k walks column k of the transposed text matrix
// (coalesced) and accumulates all views at once.  The work is tiny (<= 31 GFLOP
// per batch at 1000 classes) and stays in fp32 so the logits carry no extra
// rounding.
#include "common.h"

namespace {

constexpr int CL_THREADS = 256;
constexpr int CL_MAXT = 16;

struct ClsArgs {
    const float *feats;
    const int *row_idx;
    const float *text_t;
    int B, T, C, K;
    float scale;
    int agg, normalize;
    float *full_logits, *logits, *probs;
};

__device__ __forceinline__ float wave_red_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_red_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide reduction of T values at once; red: [4][CL_MAXT] floats
template <bool MAX>
__device__ void block_reduce(float (&v)[CL_MAXT], int T, float *red)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < CL_MAXT; t++)
        if (t < T) {
            const float r = MAX ? wave_red_max(v[t]) : wave_red_sum(v[t]);
            if (lane == 0) red[wave * CL_MAXT + t] = r;
        }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < CL_MAXT; t++)
        if (t < T) {
            const float a = red[t], b = red[CL_MAXT + t], c = red[2 * CL_MAXT + t],
                        d = red[3 * CL_MAXT + t];
            v[t] = MAX ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : (a + b) + (c + d);
        }
}

__global__ __launch_bounds__(CL_THREADS) void classify_kernel(const ClsArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *f = sm;                        // [T][C]
    float *red = sm + a.T * a.C;          // [4][CL_MAXT]
    __shared__ int s_row[CL_MAXT];
    const int b = blockIdx.x, T = a.T, C = a.C, K = a.K;

    if (threadIdx.x < T) s_row[threadIdx.x] = a.row_idx[b * T + threadIdx.x];
    __syncthreads();
    int n_valid = 0;
    for (int t = 0; t < T; t++) n_valid += s_row[t] >= 0;

    // ---- gather the view features (zeros for invalid views) ----
    for (int i = threadIdx.x; i < T * C; i += CL_THREADS) {
        const int t = i / C, c = i - t * C;
        f[i] = s_row[t] >= 0 ? a.feats[(long)s_row[t] * C + c] : 0.f;
    }
    __syncthreads();
    if (a.normalize) {                    // F.normalize(p=2, dim=-1, eps=1e-12)
        float ss[CL_MAXT];
#pragma unroll
        for (int t = 0; t < CL_MAXT; t++) {
            float s = 0.f;
            if (t < T)
                for (int c = threadIdx.x; c < C; c += CL_THREADS) s += f[t * C + c] * f[t * C + c];
            ss[t] = s;
        }
        block_reduce<false>(ss, T, red);
#pragma unroll
        for (int t = 0; t < CL_MAXT; t++)
            if (t < T) {
                const float d = fmaxf(sqrtf(ss[t]), 1e-12f);
                for (int c = threadIdx.x; c < C; c += CL_THREADS) f[t * C + c] = f[t * C + c] / d;
            }
        __syncthreads();
    }

    // ---- logits: thread k owns class k (+256, ...) for every view ----
    float *fl = a.full_logits + (long)b * T * K;
    float vmax[CL_MAXT], vsum[CL_MAXT];
#pragma unroll
    for (int t = 0; t < CL_MAXT; t++) vmax[t] = -INFINITY;
    for (int k = threadIdx.x; k < K; k += CL_THREADS) {
        float acc[CL_MAXT];
#pragma unroll
        for (int t = 0; t < CL_MAXT; t++) acc[t] = 0.f;
        for (int c = 0; c < C; c++) {
            const float w = a.text_t[(long)c * K + k];
#pragma unroll
            for (int t = 0; t < CL_MAXT; t++)
                if (t < T) acc[t] = fmaf(f[t * C + c], w, acc[t]);
        }
#pragma unroll
        for (int t = 0; t < CL_MAXT; t++)
            if (t < T) {
                const float l = s_row[t] >= 0 ? a.scale * acc[t] : 0.f;
                fl[(long)t * K + k] = l;
                vmax[t] = fmaxf(vmax[t], l);
            }
    }
    // ---- softmax statistics per view over all classes ----
    block_reduce<true>(vmax, T, red);
#pragma unroll
    for (int t = 0; t < CL_MAXT; t++) vsum[t] = 0.f;
    for (int k = threadIdx.x; k < K; k += CL_THREADS) {
#pragma unroll
        for (int t = 0; t < CL_MAXT; t++)
            if (t < T) vsum[t] += expf(fl[(long)t * K + k] - vmax[t]);
    }
    block_reduce<false>(vsum, T, red);

    // ---- aggregate over views ----
    const float nv = (float)n_valid;
    for (int k = threadIdx.x; k < K; k += CL_THREADS) {
        float lsum = 0.f, lmax = -INFINITY, psum = 0.f;
#pragma unroll
        for (int t = 0; t < CL_MAXT; t++)
            if (t < T) {
                const float l = fl[(long)t * K + k];
                const bool ok = s_row[t] >= 0;
                lsum += l;
                lmax = fmaxf(lmax, l - (ok ? 0.f : 1.f) * 1e6f);
                if (ok) psum += expf(l - vmax[t]) / vsum[t];
            }
        float lg = lsum;
        if (a.agg == EC_AGG_MEAN) lg = lsum / nv;
        if (a.agg == EC_AGG_MAX) lg = lmax;
        a.logits[(long)b * K + k] = lg;
        a.probs[(long)b * K + k] = psum / nv;
    }
}

}  // namespace

extern "C" EC_API int ec_classify(const float *feats, const int32_t *row_idx, const float *text_t,
                                  int B, int T, int C, int K, float logit_scale, int agg,
                                  int normalize, float *full_logits, float *logits, float *probs,
                                  ec_stream_t stream)
{
    EC_REQUIRE(B >= 0 && T > 0 && T <= CL_MAXT && C > 0 && K > 0,
               "ec_classify: bad shape B=%d T=%d C=%d K=%d (T <= %d)", B, T, C, K, CL_MAXT);
    EC_REQUIRE(agg == EC_AGG_SUM || agg == EC_AGG_MEAN || agg == EC_AGG_MAX,
               "ec_classify: unknown agg %d", agg);   // clip_cls.py:53
    if (B == 0) return EC_OK;
    EC_REQUIRE(feats && row_idx && text_t && full_logits && logits && probs,
               "ec_classify: null buffer");
    ClsArgs a;
    a.feats = feats, a.row_idx = row_idx, a.text_t = text_t;
    a.B = B, a.T = T, a.C = C, a.K = K, a.scale = logit_scale, a.agg = agg, a.normalize = normalize;
    a.full_logits = full_logits, a.logits = logits, a.probs = probs;
    const int lds = (T * C + 4 * CL_MAXT) * 4;
    EC_REQUIRE(lds <= 64 * 1024, "ec_classify: T*C=%d too large for LDS", T * C);
    ec::ProfScope prof(ec::PROF_CLASSIFY, static_cast<hipStream_t>(stream), 2.0 * B * T * C * K, 0);
    hipLaunchKernelGGL(classify_kernel, dim3(B), dim3(CL_THREADS), lds,
                       static_cast<hipStream_t>(stream), a);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}
