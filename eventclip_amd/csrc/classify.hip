// Logits, view aggregation and probabilities (gfx950).
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
// Round 5: the product [n_rows, C] x [C, K] runs on the matrix pipe (rounds 1 - 4: a scalar fp32 kernel, 5.8 ms for
// the 31 GFLOP of configs[4]) WITHOUT giving up fp32 accuracy -- both operands as hi + lo fp16 parts, three MFMA
// products in one ec_gemm launch (ec_gemm_args.A_lo / W_lo, EC_EPI_STORE32):
//   classify_prep_rows   feats row -> (F.normalize,) x 2^e (e per row: the row's largest element lands in [2^10, 2^11),
//                        so the lo parts are normal fp16 numbers) -> hi | lo, and the row's 2^-e
//   classify_prep_text   text_t [C, K] -> [K16, C64] hi | lo, each class row x its own 2^e'
//   ec_gemm              raw [n_rows, K16] fp32
//   classify_aggregate   per sample: logit = logit_scale 2^-e 2^-e' raw (powers of two: exact), softmax per view,
//                        aggregation over the valid views.
#include "common.h"

namespace {

constexpr int CL_THREADS = 256;
constexpr int CL_MAXT = 16;

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

// One wave per row: (L2-normalise,) scale by a power of two, split into fp16 hi + lo, columns C .. Cp - 1 zero.
__global__ __launch_bounds__(256) void classify_prep_rows(const float *feats, int n_rows, int C, int Cp, int normalize,
                                                          _Float16 *hi, _Float16 *lo, float *inv_scale)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const float *f = feats + row * C;
    float ss = 0.f, mx = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float v = f[c];
        ss = fmaf(v, v, ss), mx = fmaxf(mx, fabsf(v));
    }
    ss = wave_red_sum(ss), mx = wave_red_max(mx);
    // F.normalize(p=2, dim=-1, eps=1e-12): v / max(||v||, eps), the division as the reference does it
    const float d = normalize ? fmaxf(sqrtf(ss), 1e-12f) : 1.f;
    if (normalize) mx = mx / d;
    // 2^e with mx 2^e in [2^10, 2^11); a zero / non-finite row keeps scale 1
    int e = 0;
    if (mx > 0.f && mx < INFINITY) e = 10 - (int)floorf(log2f(mx));
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    const float sc = ldexpf(1.f, e);
    for (int c = lane; c < Cp; c += 64) {
        float v = 0.f;
        if (c < C) v = (normalize ? f[c] / d : f[c]) * sc;
        const _Float16 h = (_Float16)v;
        hi[row * Cp + c] = h;
        lo[row * Cp + c] = (_Float16)(v - (float)h);
    }
    if (lane == 0) inv_scale[row] = ldexpf(1.f, -e);
}

// text_t fp32 [C, K] -> rows k of [Kp, Cp] fp16 hi | lo, each class row scaled by its own power of two (largest element
// into [2^10, 2^11), like the feature rows: the reference's text features are unit rows, but nothing here depends on it);
// rows K .. Kp - 1 and columns C .. Cp - 1 zero.  One 256-thread workgroup per 64 classes: text_t is read with the lanes
// along k (its contiguous dimension) and the transposed rows are written with the lanes along c, through a 64 x 64 LDS
// tile (round 6; round 5 read text_t at stride K: one 4-byte element per 64-byte line).
__global__ __launch_bounds__(256) void classify_prep_text(const float *text_t, int C, int K, int Cp, int Kp, _Float16 *hi,
                                                          _Float16 *lo, float *inv_scale)
{
    __shared__ float tile[64][65];
    __shared__ float s_mx[4][64];
    __shared__ int s_e[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k0 = blockIdx.x * 64, k = k0 + lane;
    float mx = 0.f;
    if (k < K)
        for (int c = wave; c < C; c += 4) mx = fmaxf(mx, fabsf(text_t[(long)c * K + k]));
    s_mx[wave][lane] = mx;
    __syncthreads();
    if (wave == 0) {
        mx = fmaxf(fmaxf(s_mx[0][lane], s_mx[1][lane]), fmaxf(s_mx[2][lane], s_mx[3][lane]));
        int e = 0;
        if (mx > 0.f && mx < INFINITY) e = 10 - (int)floorf(log2f(mx));
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
        s_e[lane] = e;
        if (k < Kp) inv_scale[k] = ldexpf(1.f, -e);
    }
    __syncthreads();
    const int e = s_e[lane];
    for (int c0 = 0; c0 < Cp; c0 += 64) {
        for (int cc = wave; cc < 64; cc += 4) {
            const int c = c0 + cc;
            tile[cc][lane] = (k < K && c < C) ? ldexpf(text_t[(long)c * K + k], e) : 0.f;
        }
        __syncthreads();
        for (int kk = wave; kk < 64; kk += 4) {
            if (k0 + kk >= Kp) break;
            const float v = tile[lane][kk];
            const _Float16 h = (_Float16)v;
            hi[(long)(k0 + kk) * Cp + c0 + lane] = h;
            lo[(long)(k0 + kk) * Cp + c0 + lane] = (_Float16)(v - (float)h);
        }
        __syncthreads();
    }
}

struct AggArgs {
    const float *raw;         // [n_rows, Kp] fp32: (feats 2^e) . (text 2^e')
    const float *inv_scale;   // [n_rows] 2^-e of the feature rows
    const float *inv_class;   // [Kp] 2^-e of the class rows
    const int *row_idx;
    int B, T, K, Kp;
    float scale;              // logit_scale
    int agg;
    float *full_logits, *logits, *probs;
};

__global__ __launch_bounds__(CL_THREADS) void classify_aggregate_kernel(const AggArgs a)
{
    __shared__ float red[4 * CL_MAXT];
    __shared__ int s_row[CL_MAXT];
    __shared__ float s_mul[CL_MAXT];
    const int b = blockIdx.x, T = a.T, K = a.K;
    if (threadIdx.x < T) {
        const int r = a.row_idx[b * T + threadIdx.x];
        s_row[threadIdx.x] = r;
        s_mul[threadIdx.x] = r >= 0 ? a.scale * a.inv_scale[r] : 0.f;
    }
    __syncthreads();
    int n_valid = 0;
    for (int t = 0; t < T; t++) n_valid += s_row[t] >= 0;
    // ---- logits of every view (invalid views: zeros, clip_cls.py:151-152), per-view maximum ----
    float *fl = a.full_logits + (long)b * T * K;
    float vmax[CL_MAXT], vsum[CL_MAXT];
#pragma unroll
    for (int t = 0; t < CL_MAXT; t++) vmax[t] = -INFINITY;
    for (int k = threadIdx.x; k < K; k += CL_THREADS) {
#pragma unroll
        for (int t = 0; t < CL_MAXT; t++)
            if (t < T) {
                const float l = s_row[t] >= 0 ? s_mul[t] * (a.inv_class[k] * a.raw[(long)s_row[t] * a.Kp + k]) : 0.f;
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

inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }
struct ClsCarve {
    size_t a_hi, a_lo, inv, raw, total;          // per-call workspace (feature rows, raw products)
    size_t w_hi, w_lo, invk, text_total;         // prepared text planes (ec_classify_prep_text)
    int Cp, Kp;
};
ClsCarve cls_carve(int n_rows, int C, int K)
{
    ClsCarve c;
    c.Cp = (C + 63) / 64 * 64, c.Kp = (K + 15) / 16 * 16;
    size_t off = 0;
    auto take = [&](size_t b) { const size_t at = off; off += up256(b); return at; };
    c.a_hi = take((size_t)n_rows * c.Cp * 2), c.a_lo = take((size_t)n_rows * c.Cp * 2);
    c.inv = take((size_t)n_rows * 4), c.raw = take((size_t)n_rows * c.Kp * 4);
    c.total = off;
    off = 0;
    c.w_hi = take((size_t)c.Kp * c.Cp * 2), c.w_lo = take((size_t)c.Kp * c.Cp * 2), c.invk = take((size_t)c.Kp * 4);
    c.text_total = off;
    return c;
}

}  // namespace

extern "C" EC_API size_t ec_classify_v2_workspace_bytes(int n_rows, int C, int K)
{
    if (n_rows <= 0 || C <= 0 || K <= 0) return 0;
    return cls_carve(n_rows, C, K).total;
}

extern "C" EC_API size_t ec_classify_text_bytes(int C, int K)
{
    if (C <= 0 || K <= 0) return 0;
    return cls_carve(1, C, K).text_total;
}

extern "C" EC_API int ec_classify_prep_text(const float *text_t, int C, int K, void *text_ws, size_t text_ws_bytes,
                                            ec_stream_t stream)
{
    EC_REQUIRE(C > 0 && K > 0 && text_t, "ec_classify_prep_text: bad arguments C=%d K=%d", C, K);
    const ClsCarve c = cls_carve(1, C, K);
    EC_REQUIRE(text_ws && ((uintptr_t)text_ws & 255) == 0, "ec_classify_prep_text: text_ws must be 256-byte aligned");
    if (text_ws_bytes < c.text_total)
        return ec::fail(EC_ERR_WORKSPACE, "ec_classify_prep_text: text_ws %zu < %zu bytes (ec_classify_text_bytes)", text_ws_bytes,
                        c.text_total);
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned char *ws = static_cast<unsigned char *>(text_ws);
    hipLaunchKernelGGL(classify_prep_text, dim3(ec::ceil_div(c.Kp, 64)), dim3(256), 0, s, text_t, C, K, c.Cp, c.Kp,
                       reinterpret_cast<_Float16 *>(ws + c.w_hi), reinterpret_cast<_Float16 *>(ws + c.w_lo),
                       reinterpret_cast<float *>(ws + c.invk));
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

// The entry point of rounds 1 - 5 under this name took other arguments (round 4: no n_rows, no workspace; round 5: text_t
// and one workspace): a caller built against an older header gets an error code from it, never a mis-read argument list.
extern "C" EC_API int ec_classify(void)
{
    return ec::fail(EC_ERR_UNSUPPORTED, "ec_classify: removed in ABI 600 -- prepare the text planes once with ec_classify_prep_text "
                                        "and call ec_classify_v2 (include/eventclip_hip.h)");
}

extern "C" EC_API int ec_classify_v2(const float *feats, int n_rows, const int32_t *row_idx, const void *text_ws,
                                     int B, int T, int C, int K, float logit_scale, int agg,
                                     int normalize, float *full_logits, float *logits, float *probs,
                                     void *workspace, size_t workspace_bytes, ec_stream_t stream)
{
    EC_REQUIRE(B >= 0 && T > 0 && T <= CL_MAXT && C > 0 && K > 0 && n_rows >= 0,
               "ec_classify_v2: bad shape n_rows=%d B=%d T=%d C=%d K=%d (T <= %d)", n_rows, B, T, C, K, CL_MAXT);
    EC_REQUIRE(agg == EC_AGG_SUM || agg == EC_AGG_MEAN || agg == EC_AGG_MAX,
               "ec_classify_v2: unknown agg %d", agg);   // clip_cls.py:53
    if (B == 0) return EC_OK;
    EC_REQUIRE(row_idx && text_ws && full_logits && logits && probs && (feats || n_rows == 0), "ec_classify_v2: null buffer");
    EC_REQUIRE(((uintptr_t)text_ws & 255) == 0, "ec_classify_v2: text_ws must be 256-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ec::ProfScope prof(ec::PROF_CLASSIFY, s, 2.0 * n_rows * C * K, 0);
    const ClsCarve c = cls_carve(n_rows > 0 ? n_rows : 1, C, K);
    EC_REQUIRE(workspace && ((uintptr_t)workspace & 255) == 0, "ec_classify_v2: workspace must be 256-byte aligned");
    if (workspace_bytes < c.total)
        return ec::fail(EC_ERR_WORKSPACE, "ec_classify_v2: workspace %zu < %zu bytes (ec_classify_v2_workspace_bytes)", workspace_bytes, c.total);
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    const unsigned char *tw = static_cast<const unsigned char *>(text_ws);
    _Float16 *a_hi = reinterpret_cast<_Float16 *>(ws + c.a_hi), *a_lo = reinterpret_cast<_Float16 *>(ws + c.a_lo);
    const _Float16 *w_hi = reinterpret_cast<const _Float16 *>(tw + c.w_hi), *w_lo = reinterpret_cast<const _Float16 *>(tw + c.w_lo);
    const float *invk = reinterpret_cast<const float *>(tw + c.invk);
    float *inv = reinterpret_cast<float *>(ws + c.inv), *raw = reinterpret_cast<float *>(ws + c.raw);
    if (n_rows > 0) {
        hipLaunchKernelGGL(classify_prep_rows, dim3(ec::ceil_div(n_rows, 4)), dim3(256), 0, s, feats, n_rows, C, c.Cp, normalize,
                           a_hi, a_lo, inv);
        EC_CHECK_HIP(hipGetLastError());
        ec_gemm_args g = {};
        g.M = n_rows, g.N = c.Kp, g.K = c.Cp, g.dtype = EC_F16, g.epilogue = EC_EPI_STORE32, g.variant = 0;
        g.A = a_hi, g.lda = c.Cp, g.W = w_hi, g.ldw = c.Cp, g.C = raw, g.ldc = c.Kp, g.A_lo = a_lo, g.W_lo = w_lo;
        if (int rc = ec_gemm(&g, stream)) return rc;
    }
    AggArgs a;
    a.raw = raw, a.inv_scale = inv, a.inv_class = invk, a.row_idx = row_idx, a.B = B, a.T = T, a.K = K, a.Kp = c.Kp;
    a.scale = logit_scale, a.agg = agg;
    a.full_logits = full_logits, a.logits = logits, a.probs = probs;
    hipLaunchKernelGGL(classify_aggregate_kernel, dim3(B), dim3(CL_THREADS), 0, s, a);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}
