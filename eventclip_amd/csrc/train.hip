// Few-shot `text-identity` training step on cached image features (gfx950, fp32).
//
// Replaces, for adapter_type = 'text-identity' (configs/fsclip/text_adapter/*), what one optimisation
// step of the reference does around its frozen encoder:
//   forward   FSCLIPClassifier.forward, models/clip_cls.py:302-350: identity adapter, F.normalize of
//             the image features (:325-327), invalid views zeroed (:329), text_feats =
//             F.normalize(parameter) (:285-288), full_logits = logit_scale * feats @ text^T (:332),
//             aggregation (:104-129);
//   loss      calc_train_loss, :164-175: cross-entropy on the aggregated logits, or NLL of
//             log(probs + 1e-6);
//   backward  d loss / d text_feats (torch autograd upstream), written out in closed form here:
//             dL[b,v,:] per view -> dU = logit_scale * dL^T . Fn (one fp32 GEMM over the B*T views)
//             -> through the row normalisation: dT = (dU - u (u . dU)) / |t|;
//   update    torch.optim.Adam (`optimizer = 'Adam'` in the configs).
// The encoder is frozen, so its features are an input (cached once per epoch); everything here is a
// few MFLOP per sample and latency / launch bound: five small kernels on one stream.
#include "common.h"

namespace {

constexpr int TR_THREADS = 256;
constexpr int TR_WAVES = TR_THREADS / 64;

__device__ __forceinline__ float wave_sum_f(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max_f(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ float block_sum_f(float v, float *red)
{
    v = wave_sum_f(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < TR_WAVES; w++) t += red[w];
    return t;
}
__device__ float block_max_f(float v, float *red)
{
    v = wave_max_f(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = -INFINITY;
#pragma unroll
    for (int w = 0; w < TR_WAVES; w++) t = fmaxf(t, red[w]);
    return t;
}

// u[k, :] = t[k, :] / max(|t_k|, 1e-12)  (F.normalize, clip_cls.py:287); one wave per row
__global__ __launch_bounds__(64) void text_norm_kernel(const float *t, int D, float *u, float *inv_norm)
{
    const int k = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float x = t[(long)k * D + d];
        s += x * x;
    }
    const float inv = 1.f / fmaxf(__builtin_sqrtf(wave_sum_f(s)), 1e-12f);
    for (int d = lane; d < D; d += 64) u[(long)k * D + d] = t[(long)k * D + d] * inv;
    if (lane == 0) inv_norm[k] = inv;
}

// One workgroup per sample: normalised views -> Fn rows, per-view logits in LDS, loss and dL rows.
// LDS: fn [T][D] | logits [T][K] | reduction scratch
__global__ __launch_bounds__(TR_THREADS) void fs_loss_grad_kernel(
    const float *feats, const unsigned char *valid, const int *labels, const float *u, int B, int T, int D,
    int K, float scale, int agg, int probs_loss, float *Fn, float *dL, float *loss_b, float *agg_logits)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *fn = sm, *lg = sm + (size_t)T * D, *red = lg + (size_t)T * K;
    float *pvy = red + TR_WAVES;          // [T] softmax probability of the label, per view
    float *vmax = pvy + T, *vsum = vmax + T;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int y = labels[b];
    float n_valid = 0.f;
    for (int v = 0; v < T; v++) n_valid += valid[b * T + v] ? 1.f : 0.f;

    // ---- F.normalize per view, invalid views zero (clip_cls.py:325-329) ----
    for (int v = wave; v < T; v += TR_WAVES) {
        const float *f = feats + ((long)b * T + v) * D;
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s += f[d] * f[d];
        const float inv = valid[b * T + v] ? 1.f / fmaxf(__builtin_sqrtf(wave_sum_f(s)), 1e-12f) : 0.f;
        for (int d = lane; d < D; d += 64) {
            const float x = valid[b * T + v] ? f[d] * inv : 0.f;
            fn[v * D + d] = x;
            Fn[((long)b * T + v) * D + d] = x;
        }
    }
    __syncthreads();
    // ---- full_logits[v][k] = scale * fn_v . u_k (:332): one wave per class row ----
    for (int k = wave; k < K; k += TR_WAVES) {
        const float *uk = u + (long)k * D;
        for (int v0 = 0; v0 < T; v0 += 4) {
            float a[4] = {0.f, 0.f, 0.f, 0.f};
            for (int d = lane; d < D; d += 64) {
                const float w = uk[d];
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (v0 + j < T) a[j] += fn[(v0 + j) * D + d] * w;
            }
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (v0 + j < T) {
                    const float s = wave_sum_f(a[j]) * scale;
                    if (lane == 0) lg[(v0 + j) * K + k] = s;
                }
        }
    }
    __syncthreads();

    float loss = 0.f;
    if (!probs_loss) {
        // ---- aggregated logits (:104-115), softmax cross-entropy (:171) ----
        const float wv = agg == EC_AGG_MEAN ? 1.f / n_valid : 1.f;   // every view, valid or not, gets wv
        float mx = -INFINITY;
        for (int k = threadIdx.x; k < K; k += TR_THREADS) {
            float s = 0.f;
            for (int v = 0; v < T; v++) s += lg[v * K + k];
            s *= wv;
            lg[k] = s;                      // row 0 now holds the aggregated logits (each k by one thread)
            mx = fmaxf(mx, s);
        }
        mx = block_max_f(mx, red);
        float se = 0.f;
        for (int k = threadIdx.x; k < K; k += TR_THREADS) se += __expf(lg[k] - mx);
        se = block_sum_f(se, red);
        const float lse = mx + __logf(se);
        loss = lse - lg[y];
        for (int k = threadIdx.x; k < K; k += TR_THREADS) {
            const float g = (__expf(lg[k] - lse) - (k == y ? 1.f : 0.f)) / (float)B;
            if (agg_logits) agg_logits[(long)b * K + k] = lg[k];
            for (int v = 0; v < T; v++) dL[((long)b * T + v) * K + k] = g * wv;
        }
    } else {
        // ---- per-view softmax, masked mean (:123-129), NLL of log(probs + 1e-6) (:172-174) ----
        for (int v = wave; v < T; v += TR_WAVES) {
            float mx = -INFINITY;
            for (int k = lane; k < K; k += 64) mx = fmaxf(mx, lg[v * K + k]);
            mx = wave_max_f(mx);
            float se = 0.f;
            for (int k = lane; k < K; k += 64) se += __expf(lg[v * K + k] - mx);
            se = wave_sum_f(se);
            if (lane == 0) vmax[v] = mx, vsum[v] = se, pvy[v] = __expf(lg[v * K + y] - mx) / se;
        }
        __syncthreads();
        float Py = 0.f;
        for (int v = 0; v < T; v++) Py += valid[b * T + v] ? pvy[v] : 0.f;
        Py /= n_valid;
        loss = -__logf(Py + 1e-6f);
        const float dPy = -1.f / ((float)B * (Py + 1e-6f));
        for (int k = threadIdx.x; k < K; k += TR_THREADS) {
            float agg_k = 0.f;
            for (int v = 0; v < T; v++) {
                const float p = __expf(lg[v * K + k] - vmax[v]) / vsum[v];
                const float c = valid[b * T + v] ? dPy * pvy[v] / n_valid : 0.f;
                dL[((long)b * T + v) * K + k] = c * ((k == y ? 1.f : 0.f) - p);
                agg_k += lg[v * K + k];
            }
            if (agg_logits) agg_logits[(long)b * K + k] = agg == EC_AGG_MEAN ? agg_k / n_valid : agg_k;
        }
    }
    if (threadIdx.x == 0) loss_b[b] = loss;
}

// C[K, D] = alpha * A^T . Bm with A [R, K], Bm [R, D] row-major: 64 x 64 tiles, 4 x 4 per thread
__global__ __launch_bounds__(256) void sgemm_tn_kernel(const float *A, const float *Bm, int R, int K, int D,
                                                       float alpha, float *C)
{
    __shared__ float sa[16][64], sb[16][64];
    const int k0 = blockIdx.y * 64, d0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float acc[4][4] = {};
    for (int r0 = 0; r0 < R; r0 += 16) {
        for (int i = threadIdx.x; i < 16 * 64; i += 256) {
            const int r = r0 + (i >> 6), c = i & 63;
            sa[i >> 6][c] = (r < R && k0 + c < K) ? A[(long)r * K + k0 + c] : 0.f;
            sb[i >> 6][c] = (r < R && d0 + c < D) ? Bm[(long)r * D + d0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float a[4], bb[4];
#pragma unroll
            for (int i = 0; i < 4; i++) a[i] = sa[r][ty * 4 + i], bb[i] = sb[r][tx * 4 + i];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] += a[i] * bb[j];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = k0 + ty * 4 + i, d = d0 + tx * 4 + j;
            if (k < K && d < D) C[(long)k * D + d] = alpha * acc[i][j];
        }
}

// through F.normalize: dT_k = (dU_k - u_k (u_k . dU_k)) / |t_k|; block 0 also reduces the loss
__global__ __launch_bounds__(64) void text_grad_finish_kernel(const float *u, const float *dU,
                                                              const float *inv_norm, int D, const float *loss_b,
                                                              int B, float *grad, float *loss)
{
    const int k = blockIdx.x, lane = threadIdx.x;
    float dot = 0.f;
    for (int d = lane; d < D; d += 64) dot += u[(long)k * D + d] * dU[(long)k * D + d];
    dot = wave_sum_f(dot);
    const float inv = inv_norm[k];
    for (int d = lane; d < D; d += 64)
        grad[(long)k * D + d] = (dU[(long)k * D + d] - u[(long)k * D + d] * dot) * inv;
    if (k == 0) {
        float s = 0.f;
        for (int b = lane; b < B; b += 64) s += loss_b[b];
        s = wave_sum_f(s);
        if (lane == 0) loss[0] = s / (float)B;                       // F.cross_entropy / nll_loss: mean
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float *p, const float *g, float *m, float *v, long n,
                                                   float lr, float b1, float b2, float eps, float wd,
                                                   float bc1, float bc2_sqrt)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float gi = g[i];
        if (wd != 0.f) gi += wd * p[i];
        const float mi = m[i] * b1 + (1.f - b1) * gi;
        const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
        m[i] = mi, v[i] = vi;
        p[i] -= lr / bc1 * mi / (__builtin_sqrtf(vi) / bc2_sqrt + eps);
    }
}

size_t align256(size_t x) { return (x + 255) / 256 * 256; }

}  // namespace

extern "C" EC_API size_t ec_fs_text_train_workspace_bytes(int B, int T, int D, int K)
{
    if (B <= 0 || T <= 0 || D <= 0 || K <= 0) return 0;
    const size_t R = (size_t)B * T;
    return align256(R * D * 4) + align256(R * K * 4) + 2 * align256((size_t)K * D * 4) + align256((size_t)K * 4) +
           align256((size_t)B * 4);
}

extern "C" EC_API int ec_fs_text_loss_grad(const float *img_feats, const uint8_t *valid, const int32_t *labels,
                                           const float *text_param, int B, int T, int D, int K,
                                           float logit_scale, int agg, int use_probs_loss, float *loss,
                                           float *grad_text, float *agg_logits, void *workspace,
                                           size_t workspace_bytes, ec_stream_t stream)
{
    EC_REQUIRE(B > 0 && T > 0 && D > 0 && K > 0, "ec_fs_text_loss_grad: bad shape");
    EC_REQUIRE(agg == EC_AGG_SUM || agg == EC_AGG_MEAN,
               "ec_fs_text_loss_grad: agg must be sum or mean ('max' raises in the reference, clip_cls.py:117)");
    EC_REQUIRE(img_feats && valid && labels && text_param && loss && grad_text && workspace,
               "ec_fs_text_loss_grad: null buffer");
    EC_REQUIRE(workspace_bytes >= ec_fs_text_train_workspace_bytes(B, T, D, K),
               "ec_fs_text_loss_grad: workspace too small");
    const size_t lds = ((size_t)T * D + (size_t)T * K + TR_WAVES + 3 * (size_t)T) * 4;
    EC_REQUIRE(lds <= 160 * 1024, "ec_fs_text_loss_grad: T * (D + K) = %d floats exceed the LDS", T * (D + K));
    unsigned char *w = static_cast<unsigned char *>(workspace);
    const size_t R = (size_t)B * T;
    float *Fn = (float *)w;
    w += align256(R * D * 4);
    float *dL = (float *)w;
    w += align256(R * K * 4);
    float *u = (float *)w;
    w += align256((size_t)K * D * 4);
    float *dU = (float *)w;
    w += align256((size_t)K * D * 4);
    float *inv_norm = (float *)w;
    w += align256((size_t)K * 4);
    float *loss_b = (float *)w;
    hipStream_t s = static_cast<hipStream_t>(stream);
    static size_t attr_lds = 0;
    if (lds > 64 * 1024 && lds > attr_lds) {
        EC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fs_loss_grad_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    hipLaunchKernelGGL(text_norm_kernel, dim3(K), dim3(64), 0, s, text_param, D, u, inv_norm);
    hipLaunchKernelGGL(fs_loss_grad_kernel, dim3(B), dim3(TR_THREADS), lds, s, img_feats, valid, labels, u, B, T,
                       D, K, logit_scale, agg, use_probs_loss, Fn, dL, loss_b, agg_logits);
    hipLaunchKernelGGL(sgemm_tn_kernel, dim3((D + 63) / 64, (K + 63) / 64), dim3(256), 0, s, dL, Fn, (int)R, K, D,
                       logit_scale, dU);
    hipLaunchKernelGGL(text_grad_finish_kernel, dim3(K), dim3(64), 0, s, u, dU, inv_norm, D, loss_b, B,
                       grad_text, loss);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

extern "C" EC_API int ec_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                                   int64_t n, float lr, float beta1, float beta2, float eps,
                                   float weight_decay, int step, ec_stream_t stream)
{
    EC_REQUIRE(n >= 0 && step >= 1, "ec_adam_step: n=%lld step=%d", (long long)n, step);
    if (n == 0) return EC_OK;
    EC_REQUIRE(param && grad && exp_avg && exp_avg_sq, "ec_adam_step: null buffer");
    const double bc1 = 1.0 - __builtin_pow((double)beta1, (double)step);
    const double bc2 = 1.0 - __builtin_pow((double)beta2, (double)step);
    const long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), param, grad, exp_avg, exp_avg_sq, (long)n, lr, beta1,
                       beta2, eps, weight_decay, (float)bc1, (float)__builtin_sqrt(bc2));
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}
