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
#include "mfma.h"

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

// F.normalize per view, invalid views zero (clip_cls.py:325-329): one wave per view row.
// row_idx (optional): the features are compact over the valid views, row_idx[b, v] = their row
__global__ __launch_bounds__(TR_THREADS) void fs_normalize_kernel(const float *feats, const unsigned char *valid,
                                                                  const int *row_idx, int R, int D, float *Fn)
{
    const int lane = threadIdx.x & 63, r = blockIdx.x * TR_WAVES + (threadIdx.x >> 6);
    if (r >= R) return;
    const bool ok = valid[r] != 0;
    const long frow = row_idx ? (ok ? row_idx[r] : 0) : (long)r;
    const float *f = feats + frow * D;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += f[d] * f[d];
    const float inv = ok ? 1.f / fmaxf(__builtin_sqrtf(wave_sum_f(s)), 1e-12f) : 0.f;
    for (int d = lane; d < D; d += 64) Fn[(long)r * D + d] = ok ? f[d] * inv : 0.f;
}

// One workgroup per sample: its per-view logits (full_logits[v][k] = scale * fn_v . u_k, :332 -- an fp32 MFMA
// product of all view rows at once, left in the dL rows) into LDS, loss and dL rows.
// LDS: logits [T][K] | reduction scratch
__global__ __launch_bounds__(TR_THREADS) void fs_loss_grad_kernel(
    const unsigned char *valid, const int *labels, int B, int T, int K, int agg, int probs_loss, float *dL,
    float *loss_b, float *agg_logits)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *lg = sm, *red = lg + (size_t)T * K;
    float *pvy = red + TR_WAVES;          // [T] softmax probability of the label, per view
    float *vmax = pvy + T, *vsum = vmax + T;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int y = labels[b];
    float n_valid = 0.f;
    for (int v = 0; v < T; v++) n_valid += valid[b * T + v] ? 1.f : 0.f;
    for (int i = threadIdx.x; i < T * K; i += TR_THREADS) lg[i] = dL[(long)b * T * K + i];
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

// ---- generic fp32 pieces of the transformer-adapter ('text-trans') step --------------------------
// C[M, N] (row stride ldc) = alpha * sum_k A(m, k) B(k, n) + beta * C + bias[n], optional ReLU.
// A(m, k) = A[m * sam + k * sak], B(k, n) = B[k * sbk + n * sbn]: one kernel for x W^T, dy W and dy^T x.
// 64 x 64 tiles, four waves of 32 x 32, fp32 MFMA (v_mfma_f32_16x16x4_f32: full fp32 products and sums); K in
// slabs of 16 through two LDS buffers, the next slab's global loads in flight while this one multiplies (these
// products are a few hundred k-steps on a handful of workgroups: un-pipelined they ran at the memory latency,
// 104 us each).  An operand whose k index is the fast one in memory is kept [row][k] in LDS (pitch 17), the other
// way round [k][row] (pitch 80): conflict-free stores and fragment reads in both.
template <bool RELU>
__global__ __launch_bounds__(256) void sgemm_kernel(const float *A, long sam, long sak, const float *Bm, long sbk,
                                                    long sbn, int M, int N, int K, float alpha, float beta,
                                                    const float *bias, float *C, long ldc)
{
    constexpr int BK = 16, TILE = 1280;               // max(64 * 17, 16 * 80) floats
    __shared__ float sa[2][TILE], sb[2][TILE];
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c16 = lane & 15;
    const int wy = wave >> 1, wx = wave & 1;
    const bool a_kfast = sak <= sam, b_kfast = sbk < sbn;
    // element u of this thread in a slab: (k, row) with the fast index following the smaller stride
    int ak[4], am[4], bk[4], bn[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int i = threadIdx.x + 256 * u;
        if (a_kfast) ak[u] = i & 15, am[u] = i >> 4; else am[u] = i & 63, ak[u] = i >> 6;
        if (b_kfast) bk[u] = i & 15, bn[u] = i >> 4; else bn[u] = i & 63, bk[u] = i >> 6;
    }
    float ra[4], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            ra[u] = (k0 + ak[u] < K && m0 + am[u] < M) ? A[(long)(m0 + am[u]) * sam + (long)(k0 + ak[u]) * sak] : 0.f;
            rb[u] = (k0 + bk[u] < K && n0 + bn[u] < N) ? Bm[(long)(k0 + bk[u]) * sbk + (long)(n0 + bn[u]) * sbn] : 0.f;
        }
    };
    auto at = [](bool kfast, int k, int row) { return kfast ? row * 17 + k : k * 80 + row; };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            sa[buf][at(a_kfast, ak[u], am[u])] = ra[u];
            sb[buf][at(b_kfast, bk[u], bn[u])] = rb[u];
        }
    };
    ec::f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = ec::f32x4{0.f, 0.f, 0.f, 0.f};
    gload(0);
    sstore(0);
    __syncthreads();
    for (int k0 = 0, buf = 0; k0 < K; k0 += BK, buf ^= 1) {
        const bool more = k0 + BK < K;
        if (more) gload(k0 + BK);
#pragma unroll
        for (int ks = 0; ks < BK / 4; ks++) {
            const int kq = ks * 4 + g;
            float a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                a[t] = sa[buf][at(a_kfast, kq, wy * 32 + t * 16 + c16)];
                b[t] = sb[buf][at(b_kfast, kq, wx * 32 + t * 16 + c16)];
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) sstore(buf ^ 1);
        __syncthreads();
    }
    // acc[i][j][r] = row m0 + 32 wy + 16 i + 4 g + r, column n0 + 32 wx + 16 j + c16
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int n = n0 + wx * 32 + j * 16 + c16;
            if (n >= N) continue;
            const float bv = bias ? bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = m0 + wy * 32 + i * 16 + 4 * g + r;
                if (m >= M) continue;
                float v = alpha * acc[i][j][r] + bv;
                if (beta != 0.f) v += beta * C[(long)m * ldc + n];
                if (RELU) v = fmaxf(v, 0.f);
                C[(long)m * ldc + n] = v;
            }
        }
}

// out[j] = sum_r X[r, j] (* Y[r, j] when Y): bias gradients and LayerNorm gamma / beta gradients.  A workgroup
// owns 64 columns; its 256 threads are 64 columns x 4 row lanes, eight independent loads in flight per thread
// (one dependent load per row was the whole cost: R / 4 memory latencies in a row).  Fixed order: reproducible.
__global__ __launch_bounds__(256) void colsum_kernel(const float *X, const float *Y, int R, int N, float *out)
{
    __shared__ float red[4][64];
    const int j = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; u++) acc[u] = 0.f;
    if (j < N) {
        int r = part;
        for (; r + 28 < R; r += 32) {
            float x[8], y[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                x[u] = X[(long)(r + 4 * u) * N + j];
                y[u] = Y ? Y[(long)(r + 4 * u) * N + j] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) acc[u] += x[u] * y[u];
        }
        for (; r < R; r += 4) acc[0] += Y ? X[(long)r * N + j] * Y[(long)r * N + j] : X[(long)r * N + j];
    }
    red[part][threadIdx.x & 63] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (part == 0 && j < N) out[j] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// LayerNorm over rows of width N <= 1024 (eps 1e-5, torch defaults): y, xhat, rstd; one wave per row
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float *x, const float *g, const float *b, int R, int N,
                                                     float *y, float *xhat, float *rstd)
{
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= R) return;
    const float *xr = x + (long)r * N;
    float s = 0.f;
    for (int j = lane; j < N; j += 64) s += xr[j];
    const float mean = wave_sum_f(s) / (float)N;
    float q = 0.f;
    for (int j = lane; j < N; j += 64) q += (xr[j] - mean) * (xr[j] - mean);
    const float rs = 1.f / __builtin_sqrtf(wave_sum_f(q) / (float)N + 1e-5f);
    for (int j = lane; j < N; j += 64) {
        const float h = (xr[j] - mean) * rs;
        xhat[(long)r * N + j] = h;
        y[(long)r * N + j] = h * g[j] + b[j];
    }
    if (lane == 0) rstd[r] = rs;
}

// dx[r, :] (+)= rstd * (gy - mean(gy) - xhat * mean(gy * xhat)), gy = dy * gamma
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float *dy, const float *xhat, const float *rstd,
                                                     const float *g, int R, int N, float *dx, int accumulate)
{
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= R) return;
    float s1 = 0.f, s2 = 0.f;
    for (int j = lane; j < N; j += 64) {
        const float gy = dy[(long)r * N + j] * g[j];
        s1 += gy, s2 += gy * xhat[(long)r * N + j];
    }
    s1 = wave_sum_f(s1) / (float)N, s2 = wave_sum_f(s2) / (float)N;
    const float rs = rstd[r];
    for (int j = lane; j < N; j += 64) {
        const float v = rs * (dy[(long)r * N + j] * g[j] - s1 - xhat[(long)r * N + j] * s2);
        dx[(long)r * N + j] = accumulate ? dx[(long)r * N + j] + v : v;
    }
}

// Dropout (nn.TransformerEncoderLayer's p = 0.1 in train mode) without stored masks: element i of
// tensor `id` is kept iff a stateless hash of (seed, id, i) clears p; the backward pass recomputes it.
__device__ __forceinline__ bool drop_keep(unsigned long long seed, unsigned id, unsigned long long i, float p)
{
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)id + 1) + i * 0xD1B54A32D192ED03ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.f / 16777216.f) >= p;     // 24 uniform bits
}
// out = base + dropout(x) (base may be null): the residual adds after self-attention and the MLP
__global__ __launch_bounds__(256) void dropout_add_kernel(const float *base, const float *x, long n, float p,
                                                          unsigned long long seed, unsigned id, float *out)
{
    const float k = 1.f / (1.f - p);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = drop_keep(seed, id, i, p) ? x[i] * k : 0.f;
        out[i] = base ? base[i] + v : v;
    }
}
__global__ __launch_bounds__(256) void dropout_mask_kernel(long n, float p, unsigned long long seed, unsigned id,
                                                           unsigned char *mask)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        mask[i] = drop_keep(seed, id, i, p) ? 1 : 0;
}

// nn.MultiheadAttention over the T <= 16 views of a sample with src_key_padding_mask = ~valid
// (adapter.py:97-99): one 64-thread block per (sample, head).  qkv [B*T, 3d]; P [B, heads, T, T].
constexpr int AD_MAXT = 16;
__global__ __launch_bounds__(64) void adapter_attn_fwd_kernel(const float *qkv, const unsigned char *valid, int T,
                                                              int d, int heads, float *P, float *O, float drop_p,
                                                              unsigned long long seed, unsigned drop_id)
{
    __shared__ float sp[AD_MAXT][AD_MAXT];
    const int b = blockIdx.x / heads, h = blockIdx.x % heads, hd = d / heads;
    const float scale = 1.f / __builtin_sqrtf((float)hd);
    const float *base = qkv + (long)b * T * 3 * d + h * hd;
    for (int i = threadIdx.x; i < T * T; i += 64) {
        const int t = i / T, j = i % T;
        float s = 0.f;
        for (int c = 0; c < hd; c++) s += base[(long)t * 3 * d + c] * base[(long)j * 3 * d + d + c];
        sp[t][j] = valid[b * T + j] ? s * scale : -INFINITY;
    }
    __syncthreads();
    if (threadIdx.x < T) {
        const int t = threadIdx.x;
        float mx = -INFINITY, se = 0.f;
        for (int j = 0; j < T; j++) mx = fmaxf(mx, sp[t][j]);
        for (int j = 0; j < T; j++) se += __expf(sp[t][j] - mx);
        for (int j = 0; j < T; j++) {
            const float p = __expf(sp[t][j] - mx) / se;
            const long pi = (((long)b * heads + h) * T + t) * T + j;
            P[pi] = p;                                   // the softmax output (its backward needs it undropped)
            // MultiheadAttention's dropout on the attention weights
            sp[t][j] = drop_p > 0.f ? (drop_keep(seed, drop_id, pi, drop_p) ? p / (1.f - drop_p) : 0.f) : p;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < T * hd; i += 64) {
        const int t = i / hd, c = i % hd;
        float o = 0.f;
        for (int j = 0; j < T; j++) o += sp[t][j] * base[(long)j * 3 * d + 2 * d + c];
        O[((long)b * T + t) * d + h * hd + c] = o;
    }
}

__global__ __launch_bounds__(64) void adapter_attn_bwd_kernel(const float *qkv, const float *P, const float *dO,
                                                              int T, int d, int heads, float *dqkv, float drop_p,
                                                              unsigned long long seed, unsigned drop_id)
{
    __shared__ float sp[AD_MAXT][AD_MAXT], ds[AD_MAXT][AD_MAXT], sk[AD_MAXT][AD_MAXT];
    const int b = blockIdx.x / heads, h = blockIdx.x % heads, hd = d / heads;
    const float scale = 1.f / __builtin_sqrtf((float)hd);
    const float *base = qkv + (long)b * T * 3 * d + h * hd;
    const float *dob = dO + (long)b * T * d + h * hd;
    float *dq = dqkv + (long)b * T * 3 * d + h * hd;
    for (int i = threadIdx.x; i < T * T; i += 64) {
        const int t = i / T, j = i % T;
        const long pi = (((long)b * heads + h) * T + t) * T + j;
        sp[t][j] = P[pi];
        // keep-scale of the attention-weight dropout: P_drop = P * sk
        sk[t][j] = drop_p > 0.f ? (drop_keep(seed, drop_id, pi, drop_p) ? 1.f / (1.f - drop_p) : 0.f) : 1.f;
        float dp = 0.f;
        for (int c = 0; c < hd; c++) dp += dob[(long)t * d + c] * base[(long)j * 3 * d + 2 * d + c];
        ds[t][j] = dp * sk[t][j];                        // dP (through the dropout) for now
    }
    __syncthreads();
    if (threadIdx.x < T) {
        const int t = threadIdx.x;
        float dot = 0.f;
        for (int j = 0; j < T; j++) dot += sp[t][j] * ds[t][j];
        for (int j = 0; j < T; j++) ds[t][j] = sp[t][j] * (ds[t][j] - dot) * scale;   // dS (scaled)
    }
    __syncthreads();
    for (int i = threadIdx.x; i < T * hd; i += 64) {
        const int t = i / hd, c = i % hd;
        float gq = 0.f, gk = 0.f, gv = 0.f;
        for (int j = 0; j < T; j++) {
            gq += ds[t][j] * base[(long)j * 3 * d + d + c];        // dQ[t] = sum_j dS[t, j] K[j]
            gk += ds[j][t] * base[(long)j * 3 * d + c];            // dK[t] = sum_j dS[j, t] Q[j]
            gv += sp[j][t] * sk[j][t] * dob[(long)j * d + c];      // dV[t] = sum_j P_drop[j, t] dO[j]
        }
        dq[(long)t * 3 * d + c] = gq;
        dq[(long)t * 3 * d + d + c] = gk;
        dq[(long)t * 3 * d + 2 * d + c] = gv;
    }
}

// out = a * x + b * y (y may be null)
__global__ __launch_bounds__(256) void axpby_kernel(const float *x, const float *y, long n, float a, float b, float *out)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        out[i] = a * x[i] + (y ? b * y[i] : 0.f);
}
// g[i] = f[i] > 0 ? g[i] * scale : 0: backward of dropout(ReLU(.)) on the saved, already dropped
// post-activation (a dropped or clamped element is 0 there; scale = 1 / (1 - p))
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float *f, long n, float scale, float *g)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        g[i] = f[i] > 0.f ? g[i] * scale : 0.f;
}
// through F.normalize + the validity mask (clip_cls.py:325-329): dm = valid ? (dfn - fn (fn . dfn)) / |m| : 0,
// scaled by `scale` (the (1 - residual) of Adapter.residual_add on the way to out_proj)
__global__ __launch_bounds__(256) void normalize_bwd_kernel(const float *mixed, const float *fn, const float *dfn,
                                                            const unsigned char *valid, int R, int D, float scale,
                                                            float *dmixed)
{
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= R) return;
    float s = 0.f, dot = 0.f;
    for (int j = lane; j < D; j += 64) {
        const float m = mixed[(long)r * D + j];
        s += m * m, dot += fn[(long)r * D + j] * dfn[(long)r * D + j];
    }
    const float inv = valid[r] ? scale / fmaxf(__builtin_sqrtf(wave_sum_f(s)), 1e-12f) : 0.f;
    dot = wave_sum_f(dot);
    for (int j = lane; j < D; j += 64)
        dmixed[(long)r * D + j] = (dfn[(long)r * D + j] - fn[(long)r * D + j] * dot) * inv;
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
    return ec::fs_text_loss_grad(img_feats, nullptr, valid, labels, text_param, B, T, D, K, logit_scale, agg,
                                 use_probs_loss, loss, grad_text, agg_logits, workspace, workspace_bytes, stream);
}

// row_idx (optional, int32 [B, T]): img_feats holds only the valid views, row_idx[b, v] = the row of view (b, v)
int ec::fs_text_loss_grad(const float *img_feats, const int32_t *row_idx, const uint8_t *valid, const int32_t *labels,
                          const float *text_param, int B, int T, int D, int K, float logit_scale, int agg,
                          int use_probs_loss, float *loss, float *grad_text, float *agg_logits, void *workspace,
                          size_t workspace_bytes, ec_stream_t stream)
{
    EC_REQUIRE(B > 0 && T > 0 && D > 0 && K > 0, "ec_fs_text_loss_grad: bad shape");
    EC_REQUIRE(agg == EC_AGG_SUM || agg == EC_AGG_MEAN,
               "ec_fs_text_loss_grad: agg must be sum or mean ('max' raises in the reference, clip_cls.py:117)");
    EC_REQUIRE(img_feats && valid && labels && text_param && loss && grad_text && workspace,
               "ec_fs_text_loss_grad: null buffer");
    EC_REQUIRE(workspace_bytes >= ec_fs_text_train_workspace_bytes(B, T, D, K),
               "ec_fs_text_loss_grad: workspace too small");
    const size_t lds = ((size_t)T * K + TR_WAVES + 3 * (size_t)T) * 4;
    EC_REQUIRE(lds <= 160 * 1024, "ec_fs_text_loss_grad: T * K = %d floats exceed the LDS", T * K);
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
    hipLaunchKernelGGL(fs_normalize_kernel, dim3((unsigned)((R + TR_WAVES - 1) / TR_WAVES)), dim3(TR_THREADS), 0, s, img_feats, valid,
                       row_idx, (int)R, D, Fn);
    // full_logits [R, K] = logit_scale * Fn u^T, into the dL rows
    hipLaunchKernelGGL(sgemm_kernel<false>, dim3((K + 63) / 64, (unsigned)((R + 63) / 64)), dim3(256), 0, s, Fn, (long)D, 1L, u, 1L,
                       (long)D, (int)R, K, D, logit_scale, 0.f, static_cast<const float *>(nullptr), dL, (long)K);
    hipLaunchKernelGGL(fs_loss_grad_kernel, dim3(B), dim3(TR_THREADS), lds, s, valid, labels, B, T, K, agg, use_probs_loss, dL,
                       loss_b, agg_logits);
    // dU[K, D] = logit_scale * dL^T Fn: the reduction runs over the R view rows
    hipLaunchKernelGGL(sgemm_kernel<false>, dim3((D + 63) / 64, (K + 63) / 64), dim3(256), 0, s, dL, 1L, (long)K, Fn, (long)D, 1L,
                       K, D, (int)R, logit_scale, 0.f, static_cast<const float *>(nullptr), dU, (long)D);
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
    ec::ProfScope prof(ec::PROF_OPTIMIZER, static_cast<hipStream_t>(stream), 0, 28.0 * n);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), param, grad, exp_avg, exp_avg_sq, (long)n, lr, beta1,
                       beta2, eps, weight_decay, (float)bc1, (float)__builtin_sqrt(bc2));
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

// ---------------------------------------------------------------------------------------------------
// 'text-trans': TransformerAdapter (models/adapter.py:52-110) + text_feats.  Layer-wise over all
// R = B * T view rows: every nn.Linear is one sgemm_kernel launch (x W^T forward, dy W and dy^T x
// backward), the saved activations live in the workspace.  Dropout (p = 0.1 inside
// nn.TransformerEncoderLayer) is not applied: this is the deterministic function the eval-mode module
// computes, the same one the golden gradients differentiate.
// ---------------------------------------------------------------------------------------------------
namespace {

struct Carve {
    unsigned char *p;
    size_t off;
    float *f(size_t n)
    {
        float *r = p ? reinterpret_cast<float *>(p + off) : nullptr;
        off += align256(n * 4);
        return r;
    }
};

struct LayerBufs {
    float *xhat1, *rstd1, *a, *qkv, *P, *o, *h1, *xhat2, *rstd2, *bn, *f;
};

struct TransBufs {
    float *h0, *hfin, *y, *mixed, *dfn, *dy, *dh, *dh1, *dtmp_d, *dqkv, *df;
    LayerBufs L[8];
    // shared with the text-identity path
    float *Fn, *dL, *u, *dU, *inv_norm, *loss_b;
};

size_t carve_trans(Carve &c, int B, int T, int D, int K, int d, int ffn, int heads, int layers, TransBufs &t)
{
    const size_t R = (size_t)B * T;
    t.Fn = c.f(R * D), t.dL = c.f(R * K), t.u = c.f((size_t)K * D), t.dU = c.f((size_t)K * D);
    t.inv_norm = c.f(K), t.loss_b = c.f(B);
    t.h0 = c.f(R * d), t.hfin = c.f(R * d), t.y = c.f(R * D), t.mixed = c.f(R * D), t.dfn = c.f(R * D);
    t.dy = c.f(R * D), t.dh = c.f(R * d), t.dh1 = c.f(R * d), t.dtmp_d = c.f(R * d), t.dqkv = c.f(R * 3 * d);
    t.df = c.f(R * ffn);
    for (int l = 0; l < layers; l++) {
        LayerBufs &b = t.L[l];
        b.xhat1 = c.f(R * d), b.rstd1 = c.f(R), b.a = c.f(R * d), b.qkv = c.f(R * 3 * d);
        b.P = c.f((size_t)B * heads * T * T), b.o = c.f(R * d), b.h1 = c.f(R * d), b.xhat2 = c.f(R * d);
        b.rstd2 = c.f(R), b.bn = c.f(R * d), b.f = c.f(R * ffn);
    }
    return c.off;
}

void gemm(hipStream_t s, bool relu, const float *A, long sam, long sak, const float *Bm, long sbk, long sbn, int M,
          int N, int K, float alpha, float beta, const float *bias, float *C)
{
    const dim3 grid((N + 63) / 64, (M + 63) / 64);
    if (relu)
        hipLaunchKernelGGL(sgemm_kernel<true>, grid, dim3(256), 0, s, A, sam, sak, Bm, sbk, sbn, M, N, K, alpha, beta,
                           bias, C, (long)N);
    else
        hipLaunchKernelGGL(sgemm_kernel<false>, grid, dim3(256), 0, s, A, sam, sak, Bm, sbk, sbn, M, N, K, alpha, beta,
                           bias, C, (long)N);
}
// y[R, out] = x[R, in] W[out, in]^T + b (+ beta * y)
void linear_fwd(hipStream_t s, const float *x, const float *W, const float *b, int R, int in, int out, float *y,
                float beta = 0.f, bool relu = false)
{
    gemm(s, relu, x, in, 1, W, 1, in, R, out, in, 1.f, beta, b, y);
}
// dx[R, in] = dy[R, out] W (+ beta dx);  dW[out, in] = dy^T x;  db[out] = colsum(dy)
void linear_bwd(hipStream_t s, const float *x, const float *W, const float *dy, int R, int in, int out, float *dx,
                float dx_beta, float *dW, float *db)
{
    if (dx) gemm(s, false, dy, out, 1, W, in, 1, R, in, out, 1.f, dx_beta, nullptr, dx);
    gemm(s, false, dy, 1, out, x, in, 1, out, in, R, 1.f, 0.f, nullptr, dW);
    hipLaunchKernelGGL(colsum_kernel, dim3((out + 63) / 64), dim3(256), 0, s, dy, (const float *)nullptr, R, out, db);
}
unsigned blocks_for(long n) { return (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096); }

}  // namespace

extern "C" EC_API size_t ec_fs_trans_train_workspace_bytes(int B, int T, int D, int K, int d_model, int ffn_dim,
                                                           int heads, int layers)
{
    if (B <= 0 || T <= 0 || D <= 0 || K <= 0 || d_model <= 0 || ffn_dim <= 0 || layers <= 0 || layers > 8) return 0;
    Carve c{nullptr, 0};
    TransBufs t;
    return carve_trans(c, B, T, D, K, d_model, ffn_dim, heads, layers, t);
}

extern "C" EC_API int ec_fs_trans_loss_grad(const float *img_feats, const uint8_t *valid, const int32_t *labels,
                                            const float *text_param, int B, int T, int D, int K, float logit_scale,
                                            int agg, int use_probs_loss, const ec_adapter_train_params *w,
                                            const ec_adapter_train_params *g, float dropout_p,
                                            uint64_t dropout_seed, float *loss, float *grad_text,
                                            float *agg_logits, void *workspace, size_t workspace_bytes,
                                            ec_stream_t stream)
{
    EC_REQUIRE(B > 0 && T > 0 && D > 0 && K > 0, "ec_fs_trans_loss_grad: bad shape");
    EC_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "ec_fs_trans_loss_grad: dropout_p=%g", (double)dropout_p);
    EC_REQUIRE(w && g && w->blocks && g->blocks, "ec_fs_trans_loss_grad: null parameter / gradient structs");
    EC_REQUIRE(agg == EC_AGG_SUM || agg == EC_AGG_MEAN, "ec_fs_trans_loss_grad: agg must be sum or mean");
    const int d = w->d_model, ffn = w->ffn_dim, heads = w->heads, layers = w->layers;
    EC_REQUIRE(w->in_dim == D && d > 0 && d <= 1024 && heads > 0 && d % heads == 0 && layers >= 1 && layers <= 8 &&
                   T <= AD_MAXT, "ec_fs_trans_loss_grad: unsupported adapter geometry (T <= %d, layers <= 8)", AD_MAXT);
    EC_REQUIRE(img_feats && valid && labels && text_param && loss && grad_text && workspace,
               "ec_fs_trans_loss_grad: null buffer");
    Carve c{static_cast<unsigned char *>(workspace), 0};
    TransBufs t;
    const size_t need = carve_trans(c, B, T, D, K, d, ffn, heads, layers, t);
    if (need > workspace_bytes)
        return ec::fail(EC_ERR_WORKSPACE, "ec_fs_trans_loss_grad: workspace %zu < %zu bytes", workspace_bytes, need);
    const size_t lds = ((size_t)T * K + TR_WAVES + 3 * (size_t)T) * 4;
    EC_REQUIRE(lds <= 160 * 1024, "ec_fs_trans_loss_grad: T * K = %d floats exceed the LDS", T * K);
    hipStream_t s = static_cast<hipStream_t>(stream);
    static size_t attr_lds = 0;
    if (lds > 64 * 1024 && lds > attr_lds) {
        EC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fs_loss_grad_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    const int R = B * T;
    const float r = w->residual;
    const unsigned ln_grid = (unsigned)((R + 3) / 4);
    const float dp = dropout_p, keep_scale = 1.f / (1.f - dropout_p);
    const unsigned long long seed = dropout_seed;
    // dropout sites of layer l (nn.TransformerEncoderLayer, norm_first): 4 l + {0 attention weights,
    // 1 dropout1 after out_proj, 2 dropout inside the MLP, 3 dropout2 after linear2}

    // ---------------- forward (adapter.py:82-105) ----------------
    linear_fwd(s, img_feats, w->in_w, w->in_b, R, D, d, t.h0);
    const float *h = t.h0;
    for (int l = 0; l < layers; l++) {
        const ec_adapter_train_layer &p = w->blocks[l];
        LayerBufs &b = t.L[l];
        hipLaunchKernelGGL(ln_fwd_kernel, dim3(ln_grid), dim3(256), 0, s, h, p.ln1_g, p.ln1_b, R, d, b.a, b.xhat1, b.rstd1);
        linear_fwd(s, b.a, p.qkv_w, p.qkv_b, R, d, 3 * d, b.qkv);
        hipLaunchKernelGGL(adapter_attn_fwd_kernel, dim3(B * heads), dim3(64), 0, s, b.qkv, valid, T, d, heads, b.P, b.o,
                           dp, seed, (unsigned)(4 * l));
        if (dp > 0.f) {
            linear_fwd(s, b.o, p.o_w, p.o_b, R, d, d, t.dtmp_d);
            hipLaunchKernelGGL(dropout_add_kernel, dim3(blocks_for((long)R * d)), dim3(256), 0, s, h, t.dtmp_d,
                               (long)R * d, dp, seed, (unsigned)(4 * l + 1), b.h1);   // h1 = h + dropout1(out_proj(o))
        } else {
            hipLaunchKernelGGL(axpby_kernel, dim3(blocks_for((long)R * d)), dim3(256), 0, s, h, (const float *)nullptr,
                               (long)R * d, 1.f, 0.f, b.h1);
            linear_fwd(s, b.o, p.o_w, p.o_b, R, d, d, b.h1, 1.f);                   // h1 = h + out_proj(o)
        }
        hipLaunchKernelGGL(ln_fwd_kernel, dim3(ln_grid), dim3(256), 0, s, b.h1, p.ln2_g, p.ln2_b, R, d, b.bn, b.xhat2,
                           b.rstd2);
        linear_fwd(s, b.bn, p.w1, p.b1, R, d, ffn, b.f, 0.f, true);                   // relu(linear1)
        if (dp > 0.f)
            hipLaunchKernelGGL(dropout_add_kernel, dim3(blocks_for((long)R * ffn)), dim3(256), 0, s,
                               (const float *)nullptr, b.f, (long)R * ffn, dp, seed, (unsigned)(4 * l + 2), b.f);
        if (dp > 0.f) {
            linear_fwd(s, b.f, p.w2, p.b2, R, ffn, d, t.dtmp_d);
            hipLaunchKernelGGL(dropout_add_kernel, dim3(blocks_for((long)R * d)), dim3(256), 0, s, b.h1, t.dtmp_d,
                               (long)R * d, dp, seed, (unsigned)(4 * l + 3), t.hfin);  // h = h1 + dropout2(linear2(f))
        } else {
            hipLaunchKernelGGL(axpby_kernel, dim3(blocks_for((long)R * d)), dim3(256), 0, s, b.h1, (const float *)nullptr,
                               (long)R * d, 1.f, 0.f, t.hfin);
            linear_fwd(s, b.f, p.w2, p.b2, R, ffn, d, t.hfin, 1.f);                   // h = h1 + linear2(f)
        }
        h = t.hfin;      // this layer's input is dead (ln_1 consumed it, h1 took its copy): reuse its slot
    }
    linear_fwd(s, h, w->out_w, w->out_b, R, d, D, t.y);
    hipLaunchKernelGGL(axpby_kernel, dim3(blocks_for((long)R * D)), dim3(256), 0, s, img_feats, t.y, (long)R * D, r,
                       1.f - r, t.mixed);                                           // Adapter.residual_add

    // ---------------- loss, dL, text gradient (shared with 'text-identity') ----------------
    hipLaunchKernelGGL(text_norm_kernel, dim3(K), dim3(64), 0, s, text_param, D, t.u, t.inv_norm);
    hipLaunchKernelGGL(fs_normalize_kernel, dim3((unsigned)((R + TR_WAVES - 1) / TR_WAVES)), dim3(TR_THREADS), 0, s, t.mixed, valid,
                       (const int *)nullptr, R, D, t.Fn);
    hipLaunchKernelGGL(sgemm_kernel<false>, dim3((K + 63) / 64, (unsigned)((R + 63) / 64)), dim3(256), 0, s, t.Fn, (long)D, 1L, t.u,
                       1L, (long)D, R, K, D, logit_scale, 0.f, static_cast<const float *>(nullptr), t.dL, (long)K);
    hipLaunchKernelGGL(fs_loss_grad_kernel, dim3(B), dim3(TR_THREADS), lds, s, valid, labels, B, T, K, agg, use_probs_loss, t.dL,
                       t.loss_b, agg_logits);
    hipLaunchKernelGGL(sgemm_kernel<false>, dim3((D + 63) / 64, (K + 63) / 64), dim3(256), 0, s, t.dL, 1L, (long)K, t.Fn, (long)D,
                       1L, K, D, R, logit_scale, 0.f, static_cast<const float *>(nullptr), t.dU, (long)D);
    hipLaunchKernelGGL(text_grad_finish_kernel, dim3(K), dim3(64), 0, s, t.u, t.dU, t.inv_norm, D, t.loss_b, B,
                       grad_text, loss);

    // ---------------- backward into the adapter ----------------
    gemm(s, false, t.dL, K, 1, t.u, D, 1, R, D, K, logit_scale, 0.f, nullptr, t.dfn);   // dFn = s dL u
    hipLaunchKernelGGL(normalize_bwd_kernel, dim3(ln_grid), dim3(256), 0, s, t.mixed, t.Fn, t.dfn, valid, R, D, 1.f - r,
                       t.dy);                                                        // dY = (1 - r) dMixed
    linear_bwd(s, h, w->out_w, t.dy, R, d, D, t.dh, 0.f, g->out_w, g->out_b);
    for (int l = layers - 1; l >= 0; l--) {
        const ec_adapter_train_layer &p = w->blocks[l];
        const ec_adapter_train_layer &q = g->blocks[l];
        LayerBufs &b = t.L[l];
        // h = h1 + dropout2(linear2(dropout(relu(linear1(ln2(h1))))))
        const float *dlin2 = t.dh;
        if (dp > 0.f) {
            hipLaunchKernelGGL(dropout_add_kernel, dim3(blocks_for((long)R * d)), dim3(256), 0, s,
                               (const float *)nullptr, t.dh, (long)R * d, dp, seed, (unsigned)(4 * l + 3), t.dtmp_d);
            dlin2 = t.dtmp_d;
        }
        linear_bwd(s, b.f, p.w2, dlin2, R, ffn, d, t.df, 0.f, const_cast<float *>(q.w2), const_cast<float *>(q.b2));
        hipLaunchKernelGGL(relu_bwd_kernel, dim3(blocks_for((long)R * ffn)), dim3(256), 0, s, b.f, (long)R * ffn,
                           keep_scale, t.df);
        linear_bwd(s, b.bn, p.w1, t.df, R, d, ffn, t.dtmp_d, 0.f, const_cast<float *>(q.w1), const_cast<float *>(q.b1));
        hipLaunchKernelGGL(colsum_kernel, dim3((d + 63) / 64), dim3(256), 0, s, t.dtmp_d, b.xhat2, R, d,
                           const_cast<float *>(q.ln2_g));
        hipLaunchKernelGGL(colsum_kernel, dim3((d + 63) / 64), dim3(256), 0, s, t.dtmp_d, (const float *)nullptr, R, d,
                           const_cast<float *>(q.ln2_b));
        hipLaunchKernelGGL(axpby_kernel, dim3(blocks_for((long)R * d)), dim3(256), 0, s, t.dh, (const float *)nullptr,
                           (long)R * d, 1.f, 0.f, t.dh1);
        hipLaunchKernelGGL(ln_bwd_kernel, dim3(ln_grid), dim3(256), 0, s, t.dtmp_d, b.xhat2, b.rstd2, p.ln2_g, R, d, t.dh1,
                           1);
        // h1 = h + dropout1(out_proj(attn(in_proj(ln1(h)))))
        const float *dlin_o = t.dh1;
        if (dp > 0.f) {
            hipLaunchKernelGGL(dropout_add_kernel, dim3(blocks_for((long)R * d)), dim3(256), 0, s,
                               (const float *)nullptr, t.dh1, (long)R * d, dp, seed, (unsigned)(4 * l + 1), t.dh);
            dlin_o = t.dh;        // dh is rebuilt from dh1 below
        }
        linear_bwd(s, b.o, p.o_w, dlin_o, R, d, d, t.dtmp_d, 0.f, const_cast<float *>(q.o_w), const_cast<float *>(q.o_b));
        hipLaunchKernelGGL(adapter_attn_bwd_kernel, dim3(B * heads), dim3(64), 0, s, b.qkv, b.P, t.dtmp_d, T, d, heads,
                           t.dqkv, dp, seed, (unsigned)(4 * l));
        linear_bwd(s, b.a, p.qkv_w, t.dqkv, R, d, 3 * d, t.dtmp_d, 0.f, const_cast<float *>(q.qkv_w),
                   const_cast<float *>(q.qkv_b));
        hipLaunchKernelGGL(colsum_kernel, dim3((d + 63) / 64), dim3(256), 0, s, t.dtmp_d, b.xhat1, R, d,
                           const_cast<float *>(q.ln1_g));
        hipLaunchKernelGGL(colsum_kernel, dim3((d + 63) / 64), dim3(256), 0, s, t.dtmp_d, (const float *)nullptr, R, d,
                           const_cast<float *>(q.ln1_b));
        hipLaunchKernelGGL(axpby_kernel, dim3(blocks_for((long)R * d)), dim3(256), 0, s, t.dh1, (const float *)nullptr,
                           (long)R * d, 1.f, 0.f, t.dh);
        hipLaunchKernelGGL(ln_bwd_kernel, dim3(ln_grid), dim3(256), 0, s, t.dtmp_d, b.xhat1, b.rstd1, p.ln1_g, R, d, t.dh, 1);
    }
    linear_bwd(s, img_feats, w->in_w, t.dh, R, D, d, nullptr, 0.f, g->in_w, g->in_b);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

extern "C" EC_API int ec_sgemm(const float *A, long sam, long sak, const float *B, long sbk, long sbn, int M, int N,
                               int K, float alpha, float beta, float *C, long ldc, ec_stream_t stream)
{
    EC_REQUIRE(M >= 0 && N >= 0 && K >= 0 && ldc >= N, "ec_sgemm: bad shape %d x %d x %d (ldc %ld)", M, N, K, ldc);
    if (M == 0 || N == 0) return EC_OK;
    EC_REQUIRE(A && B && C, "ec_sgemm: null buffer");
    const dim3 grid((N + 63) / 64, (M + 63) / 64);
    ec::ProfScope prof(ec::PROF_SGEMM, static_cast<hipStream_t>(stream), 2.0 * M * N * K, 0);
    hipLaunchKernelGGL(sgemm_kernel<false>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), A, sam, sak, B, sbk,
                       sbn, M, N, K, alpha, beta, (const float *)nullptr, C, ldc);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

extern "C" EC_API int ec_dropout_mask(uint64_t seed, uint32_t site, int64_t n, float p, uint8_t *mask,
                                      ec_stream_t stream)
{
    EC_REQUIRE(n >= 0 && p >= 0.f && p < 1.f, "ec_dropout_mask: n=%lld p=%g", (long long)n, (double)p);
    if (n == 0) return EC_OK;
    EC_REQUIRE(mask, "ec_dropout_mask: null buffer");
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(blocks_for((long)n)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       (long)n, p, (unsigned long long)seed, (unsigned)site, mask);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}
