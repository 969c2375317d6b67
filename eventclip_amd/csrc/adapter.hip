// Few-shot feature adapter forward (gfx950, fp32).
//
// Replaces TransformerAdapter.forward + Adapter.residual_add of the reference
// (models/adapter.py:82-105, :22-25): in_proj Linear(D -> d_model), L x pre-LN
// nn.TransformerEncoderLayer (torch defaults: ReLU, LayerNorm eps 1e-5, no final
// norm, dropout off in eval) with src_key_padding_mask = ~valid, out_proj
// Linear(d_model -> D), then  in * r + new * (1 - r).
//
// The reference issues ~25 tiny fp32 kernels over [B, T <= 10, 256] tensors and is
// launch-latency bound.  Here one 256-thread workgroup owns one sample and walks
// the whole adapter with every activation in LDS: thread n owns output column n of
// each Linear for all T tokens at once (weights are pre-transposed to [K, N] so a
// wave reads 256 contiguous bytes per k), LayerNorm is one wave per token, and the
// T x T attention is done with scalar loops.  fp32 throughout, as the reference's
// adapter (clip_cls.py:281-288).
#include "common.h"

namespace {

constexpr int AD_THREADS = 256;
constexpr int AD_MAXT = 16;
constexpr int AD_MAXL = 4;

struct AdArgs {
    int in_dim, dm, heads, ffn, layers, B, T;
    float residual;
    const float *in_w_t, *in_b, *out_w_t, *out_b;
    ec_adapter_layer L[AD_MAXL];
    const float *feats;
    const int *row_idx;
    float *out;
};

// out[t][n] (op)= sum_k in[t][k] * Wt[k][n] + bias[n]; K % 4 == 0.
// mode 0: store, 1: relu-store, 2: accumulate into out.
__device__ void linear(const float *in, int ld_in, int K, const float *__restrict__ Wt,
                       const float *__restrict__ bias, int N, int T, float *out, int ld_out, int mode)
{
    for (int n = threadIdx.x; n < N; n += AD_THREADS) {
        float acc[AD_MAXT];
        const float b = bias[n];
#pragma unroll
        for (int t = 0; t < AD_MAXT; t++) acc[t] = b;
        for (int k = 0; k < K; k += 4) {
            const float w0 = Wt[(long)k * N + n], w1 = Wt[(long)(k + 1) * N + n],
                        w2 = Wt[(long)(k + 2) * N + n], w3 = Wt[(long)(k + 3) * N + n];
#pragma unroll
            for (int t = 0; t < AD_MAXT; t++)
                if (t < T) {
                    const float4 v = *reinterpret_cast<const float4 *>(in + t * ld_in + k);
                    acc[t] = fmaf(v.x, w0, acc[t]);
                    acc[t] = fmaf(v.y, w1, acc[t]);
                    acc[t] = fmaf(v.z, w2, acc[t]);
                    acc[t] = fmaf(v.w, w3, acc[t]);
                }
        }
#pragma unroll
        for (int t = 0; t < AD_MAXT; t++)
            if (t < T) {
                float v = acc[t];
                if (mode == 1) v = fmaxf(v, 0.f);
                if (mode == 2) v += out[t * ld_out + n];
                out[t * ld_out + n] = v;
            }
    }
    __syncthreads();
}

__device__ void layer_norm(const float *in, float *out, int T, int W, const float *__restrict__ g,
                           const float *__restrict__ b)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = wave; t < T; t += AD_THREADS / 64) {
        float s = 0.f;
        for (int c = lane; c < W; c += 64) s += in[t * W + c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = s / (float)W;
        float q = 0.f;
        for (int c = lane; c < W; c += 64) {
            const float d = in[t * W + c] - mean;
            q += d * d;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        const float rstd = 1.f / sqrtf(q / (float)W + 1e-5f);
        for (int c = lane; c < W; c += 64) out[t * W + c] = (in[t * W + c] - mean) * rstd * g[c] + b[c];
    }
    __syncthreads();
}

__global__ __launch_bounds__(AD_THREADS) void adapter_kernel(const AdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int T = a.T, D = a.in_dim, dm = a.dm, hd = a.dm / a.heads;
    const int wide = max(3 * dm, a.ffn);
    float *x0 = sm;                      // [T][D]   adapter input (zero rows for padded views)
    float *h = x0 + T * D;               // [T][dm]  residual stream
    float *y = h + T * dm;               // [T][dm]  LayerNorm output / attention output
    float *big = y + T * dm;             // [T][wide] qkv or FFN hidden
    float *sc = big + T * wide;          // [heads][T][T] attention probabilities
    __shared__ int s_row[AD_MAXT];

    const int b = blockIdx.x;
    if (threadIdx.x < T) s_row[threadIdx.x] = a.row_idx[b * T + threadIdx.x];
    __syncthreads();
    for (int i = threadIdx.x; i < T * D; i += AD_THREADS) {
        const int t = i / D, c = i - t * D;
        x0[i] = s_row[t] >= 0 ? a.feats[(long)s_row[t] * D + c] : 0.f;   // clip_cls.py:319-321
    }
    __syncthreads();

    linear(x0, D, D, a.in_w_t, a.in_b, dm, T, h, dm, 0);                  // adapter.py:95
    for (int l = 0; l < a.layers; l++) {
        const ec_adapter_layer &w = a.L[l];
        // x = x + SA(norm1(x)) with key padding mask
        layer_norm(h, y, T, dm, w.ln1_g, w.ln1_b);
        linear(y, dm, dm, w.qkv_w_t, w.qkv_b, 3 * dm, T, big, wide, 0);
        const float scale = rsqrtf((float)hd);
        for (int i = threadIdx.x; i < a.heads * T * T; i += AD_THREADS) {
            const int hh = i / (T * T), r = i - hh * T * T, tq = r / T, tk = r - tq * T;
            const float *q = big + tq * wide + hh * hd, *k = big + tk * wide + dm + hh * hd;
            float s = 0.f;
            for (int d = 0; d < hd; d++) s = fmaf(q[d] * scale, k[d], s);
            sc[i] = s_row[tk] >= 0 ? s : -INFINITY;                        // adapter.py:98-99
        }
        __syncthreads();
        for (int i = threadIdx.x; i < a.heads * T; i += AD_THREADS) {
            float *row = sc + i * T;
            float m = -INFINITY;
            for (int t = 0; t < T; t++) m = fmaxf(m, row[t]);
            float s = 0.f;
            for (int t = 0; t < T; t++) {
                row[t] = expf(row[t] - m);
                s += row[t];
            }
            for (int t = 0; t < T; t++) row[t] /= s;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < T * dm; i += AD_THREADS) {
            const int tq = i / dm, c = i - tq * dm, hh = c / hd;
            const float *p = sc + (hh * T + tq) * T;
            float o = 0.f;
            for (int tk = 0; tk < T; tk++) o = fmaf(p[tk], big[tk * wide + 2 * dm + c], o);
            y[i] = o;
        }
        __syncthreads();
        linear(y, dm, dm, w.o_w_t, w.o_b, dm, T, h, dm, 2);
        // x = x + FF(norm2(x)), ReLU
        layer_norm(h, y, T, dm, w.ln2_g, w.ln2_b);
        linear(y, dm, dm, w.w1_t, w.b1, a.ffn, T, big, wide, 1);
        linear(big, wide, a.ffn, w.w2_t, w.b2, dm, T, h, dm, 2);
    }
    // out_proj, then in * r + new * (1 - r)   (adapter.py:102-105, :22-25)
    const float r = a.residual;
    float *dst = a.out + (long)b * T * D;
    for (int n = threadIdx.x; n < D; n += AD_THREADS) {
        float acc[AD_MAXT];
        const float bb = a.out_b[n];
#pragma unroll
        for (int t = 0; t < AD_MAXT; t++) acc[t] = bb;
        for (int k = 0; k < dm; k += 4) {
            const float w0 = a.out_w_t[(long)k * D + n], w1 = a.out_w_t[(long)(k + 1) * D + n],
                        w2 = a.out_w_t[(long)(k + 2) * D + n], w3 = a.out_w_t[(long)(k + 3) * D + n];
#pragma unroll
            for (int t = 0; t < AD_MAXT; t++)
                if (t < T) {
                    const float4 v = *reinterpret_cast<const float4 *>(h + t * dm + k);
                    acc[t] = fmaf(v.x, w0, acc[t]);
                    acc[t] = fmaf(v.y, w1, acc[t]);
                    acc[t] = fmaf(v.z, w2, acc[t]);
                    acc[t] = fmaf(v.w, w3, acc[t]);
                }
        }
#pragma unroll
        for (int t = 0; t < AD_MAXT; t++)
            if (t < T) dst[(long)t * D + n] = x0[t * D + n] * r + acc[t] * (1.f - r);
    }
}

}  // namespace

extern "C" EC_API int ec_adapter_forward(const ec_adapter_weights *w, const float *feats,
                                         const int32_t *row_idx, int B, int T, float *out,
                                         ec_stream_t stream)
{
    EC_REQUIRE(w && w->layer, "ec_adapter_forward: weights are null");
    EC_REQUIRE(B >= 0 && T > 0 && T <= AD_MAXT, "ec_adapter_forward: T=%d (<= %d)", T, AD_MAXT);
    EC_REQUIRE(w->layers >= 0 && w->layers <= AD_MAXL, "ec_adapter_forward: %d layers (<= %d)",
               w->layers, AD_MAXL);
    EC_REQUIRE(w->in_dim % 4 == 0 && w->d_model % 4 == 0 && w->ffn % 4 == 0 && w->heads > 0 &&
                   w->d_model % w->heads == 0,
               "ec_adapter_forward: bad dims");
    EC_REQUIRE(w->residual >= 0.f && w->residual <= 1.f, "ec_adapter_forward: residual %f",
               w->residual);   // adapter.py:17-18
    if (B == 0) return EC_OK;
    EC_REQUIRE(feats && row_idx && out, "ec_adapter_forward: null buffer");
    AdArgs a;
    a.in_dim = w->in_dim, a.dm = w->d_model, a.heads = w->heads, a.ffn = w->ffn, a.layers = w->layers;
    a.B = B, a.T = T, a.residual = w->residual;
    a.in_w_t = w->in_w_t, a.in_b = w->in_b, a.out_w_t = w->out_w_t, a.out_b = w->out_b;
    for (int l = 0; l < w->layers; l++) a.L[l] = w->layer[l];
    a.feats = feats, a.row_idx = row_idx, a.out = out;
    const int wide = (3 * a.dm > a.ffn) ? 3 * a.dm : a.ffn;
    const int lds = (T * a.in_dim + 2 * T * a.dm + T * wide + a.heads * T * T) * 4;
    EC_REQUIRE(lds <= 160 * 1024, "ec_adapter_forward: needs %d bytes of LDS", lds);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (lds > 64 * 1024)
        EC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(adapter_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    // flops per sample: 2*T*(D*dm*2 + layers*(4*dm*dm + 2*dm*ffn)) + attention
    const double fl = 2.0 * T * (2.0 * a.in_dim * a.dm + a.layers * (4.0 * a.dm * a.dm + 2.0 * a.dm * a.ffn));
    ec::ProfScope prof(ec::PROF_ADAPTER, s, fl * B, 0);
    hipLaunchKernelGGL(adapter_kernel, dim3(B), dim3(AD_THREADS), lds, s, a);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}
