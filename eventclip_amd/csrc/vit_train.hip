// Fine-tuning the CLIP vision tower on the MI355X: forward with a tape, backward, weight packing.
//
// Replaces what torch autograd does for the reference's FTCLIPClassifier (models/clip_cls_ft.py:
// _build_clip :44-80 marks model.visual -- or sub-sets, or LoRA factors acting through merged weights,
// models/lora.py:138-150 -- trainable; forward :196-243 calls encode_image :180-183; loss :245-256) around
// un-vendored openai/CLIP's VisionTransformer (conv1 -> class token + positions -> ln_pre -> L x
// {x += MHA(ln_1 x); x += c_proj(QuickGELU(c_fc(ln_2 x)))} -> ln_post(CLS) @ proj).
//
// Layout of a step (M = n_img * S token rows, W = width):
//   forward   the inference kernels, keeping per block both residual-stream inputs (fp32), both LayerNorm outputs,
//             q | k | v, the attention output with its log-sum-exp, the MLP pre-activation and its QuickGELU
//             (16 bit): 36 M W bytes;
//   backward  the residual-stream gradient dx stays fp32.  Every nn.Linear is two 16-bit MFMA GEMMs:
//               dX = dY W        ec_gemm with the weight's transposed copy as the [N, K] operand;
//               dW = dY^T X      ec_gemm over the ROW index of the two row-major activations as the passes left
//                                them (ec_gemm_args.transposed: tiles read column-major out of LDS), the row
//                                dimension cut into K-batches (`splits`) so that the few 256 x 256 output
//                                tiles of a weight gradient still fill 256 CUs; the partial sums are reduced
//                                in fp32 in a fixed order (no atomics: reproducible).  Only the patch
//                                embedding's gradient still goes through transposed copies (its rows skip
//                                the class tokens);
//             the QuickGELU derivative is fused into the GEMM that produces it (EC_EPI_GELU_BWD16), LayerNorm
//             backward recomputes its statistics from the saved input, attention backward recomputes the
//             probabilities from the saved log-sum-exp (attention_bwd.hip);
//   gradients come out fp32 in the state dict's layouts; a NULL pointer skips the work behind it
//   (LoRA on q k v o needs the attention weight gradients only: none of the MLP's dW GEMMs run).
#include "common.h"
#include "mfma.h"
#include "tower_ops.h"

namespace {

using namespace ec;
using namespace ec_tower;

constexpr int LN_MAXV = 8;   // float4 per lane: width <= 2048
constexpr int SIZING_CUS = 256;

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int units(int width, int lane)
{
    const int total = width / 4;
    return (total - lane + 63) / 64;
}
__device__ __forceinline__ float quick_gelu_f(float x)
{
    return x * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * x));   // as the GEMM epilogue computes it
}

// ------------------------------------------------------------------------------------------
// Transposed 16-bit copies: dst_t[n][m] = f(src[row(m)][n]) for m < M, 0 for M <= m < Mp.
// MODE 0: 16-bit source; 1: QuickGELU of a 16-bit source; 2: fp32 source, optionally also written row-major
// as 16 bit (the dX GEMM's operand).  (The blocks' weight gradients no longer need these copies: weight_grad_rows.)
// Logical row m reads source row (m / seq_out) * seq_in + seq_off + m % seq_out (seq_out = 0: row m):
// the patch rows of the embedding gradient skip every sequence's class-token row.
// ------------------------------------------------------------------------------------------
template <int DT, int MODE>
__global__ __launch_bounds__(256) void transpose_kernel(const void *src, long ld, int M, int N, int Mp, int seq_out,
                                                        int seq_in, int seq_off, void *dst_t, void *dst_rm)
{
    typedef typename T16<DT>::elem elem;
    typedef typename T16<DT>::v8 v8;
    constexpr int PITCH = 72;   // elements: 144-B rows, 16-B aligned
    __shared__ __attribute__((aligned(16))) elem tile[64 * PITCH];
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const int ch = threadIdx.x & 7;
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int ml = (threadIdx.x >> 3) + 32 * it;
        const int m = m0 + ml, n = n0 + ch * 8;
        v8 v;
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = (elem)0.f;
        if (m < M && n < N) {
            const long row = seq_out ? (long)(m / seq_out) * seq_in + seq_off + m % seq_out : m;
            if (MODE == 2) {
                const float *p = (const float *)src + row * ld + n;
                const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
                v[0] = to16(a.x, elem()), v[1] = to16(a.y, elem()), v[2] = to16(a.z, elem()), v[3] = to16(a.w, elem());
                v[4] = to16(b.x, elem()), v[5] = to16(b.y, elem()), v[6] = to16(b.z, elem()), v[7] = to16(b.w, elem());
                if (dst_rm) *reinterpret_cast<v8 *>((elem *)dst_rm + (long)m * N + n) = v;
            } else {
                v = *reinterpret_cast<const v8 *>((const elem *)src + row * ld + n);
                if (MODE == 1) {
#pragma unroll
                    for (int j = 0; j < 8; j++) v[j] = to16(quick_gelu_f((float)v[j]), elem());
                }
            }
        }
        if (dst_t) *reinterpret_cast<v8 *>(&tile[ml * PITCH + ch * 8]) = v;
    }
    if (!dst_t) return;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int nl = (threadIdx.x >> 3) + 32 * it;
        if (n0 + nl >= N) continue;
        v8 v;
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = tile[(ch * 8 + j) * PITCH + nl];
        *reinterpret_cast<v8 *>((elem *)dst_t + (long)(n0 + nl) * Mp + m0 + ch * 8) = v;
    }
}

// ------------------------------------------------------------------------------------------
// LayerNorm backward, one wave per row, statistics recomputed from the saved input:
//   xh = (x - mean) rstd, g = dy gamma, dx = rstd (g - mean(g) - xh mean(g xh)).
// d_gamma / d_beta: per-lane column sums over the rows a workgroup walks, reduced over its 4 waves in
// LDS, one partial row pair per workgroup (summed by reduce_kernel in a fixed order).
// ------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float *x, long ldx, const float *dy, long ldy,
                                                     const float *gamma, int rows, int width, float eps, float *dx,
                                                     long ldo, int accumulate, float *partials, void *dx16)
{
    typedef typename T16<DT>::elem elem;
    typedef typename T16<DT>::v4 v4;
    __shared__ float red[3][2 * LN_MAXV * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = units(width, lane);
    float4 sg[LN_MAXV], sb[LN_MAXV], gm[LN_MAXV];
#pragma unroll
    for (int i = 0; i < LN_MAXV; i++) {
        sg[i] = sb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        gm[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < nv) gm[i] = *reinterpret_cast<const float4 *>(gamma + (i * 64 + lane) * 4);
    }
    const float inv_w = 1.f / (float)width;
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        float4 v[LN_MAXV], d[LN_MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; i++)
            if (i < nv) {
                v[i] = *reinterpret_cast<const float4 *>(x + row * ldx + (i * 64 + lane) * 4);
                d[i] = *reinterpret_cast<const float4 *>(dy + row * ldy + (i * 64 + lane) * 4);
                s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            }
        const float mean = wave_sum(s) * inv_w;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; i++)
            if (i < nv) {
                v[i].x -= mean, v[i].y -= mean, v[i].z -= mean, v[i].w -= mean;
                q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
            }
        const float rstd = 1.f / __builtin_sqrtf(wave_sum(q) * inv_w + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; i++)
            if (i < nv) {
                v[i].x *= rstd, v[i].y *= rstd, v[i].z *= rstd, v[i].w *= rstd;      // x hat
                sg[i].x += d[i].x * v[i].x, sg[i].y += d[i].y * v[i].y;
                sg[i].z += d[i].z * v[i].z, sg[i].w += d[i].w * v[i].w;
                sb[i].x += d[i].x, sb[i].y += d[i].y, sb[i].z += d[i].z, sb[i].w += d[i].w;
                d[i].x *= gm[i].x, d[i].y *= gm[i].y, d[i].z *= gm[i].z, d[i].w *= gm[i].w;   // g
                s1 += (d[i].x + d[i].y) + (d[i].z + d[i].w);
                s2 += (d[i].x * v[i].x + d[i].y * v[i].y) + (d[i].z * v[i].z + d[i].w * v[i].w);
            }
        s1 = wave_sum(s1) * inv_w, s2 = wave_sum(s2) * inv_w;
#pragma unroll
        for (int i = 0; i < LN_MAXV; i++)
            if (i < nv) {
                float *o = dx + row * ldo + (i * 64 + lane) * 4;
                float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
                if (accumulate) r = *reinterpret_cast<const float4 *>(o);
                r.x += rstd * (d[i].x - s1 - v[i].x * s2);
                r.y += rstd * (d[i].y - s1 - v[i].y * s2);
                r.z += rstd * (d[i].z - s1 - v[i].z * s2);
                r.w += rstd * (d[i].w - s1 - v[i].w * s2);
                *reinterpret_cast<float4 *>(o) = r;
                if (dx16) {   // the 16-bit copy the next dX GEMM reads (same row stride)
                    const v4 h = {to16(r.x, elem()), to16(r.y, elem()), to16(r.z, elem()), to16(r.w, elem())};
                    *reinterpret_cast<v4 *>((elem *)dx16 + row * ldo + (i * 64 + lane) * 4) = h;
                }
            }
    }
    if (!partials) return;
    // waves 1..3 hand their sums to wave 0 through LDS
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < LN_MAXV; i++)
            if (i < nv) {
                *reinterpret_cast<float4 *>(&red[wave - 1][(i * 64 + lane) * 4]) = sg[i];
                *reinterpret_cast<float4 *>(&red[wave - 1][LN_MAXV * 256 + (i * 64 + lane) * 4]) = sb[i];
            }
    }
    __syncthreads();
    if (wave == 0) {
        float *pg = partials + (long)blockIdx.x * 2 * width, *pb = pg + width;
#pragma unroll
        for (int i = 0; i < LN_MAXV; i++)
            if (i < nv) {
                float4 a = sg[i], b = sb[i];
#pragma unroll
                for (int w = 0; w < 3; w++) {
                    const float4 ra = *reinterpret_cast<const float4 *>(&red[w][(i * 64 + lane) * 4]);
                    const float4 rb = *reinterpret_cast<const float4 *>(&red[w][LN_MAXV * 256 + (i * 64 + lane) * 4]);
                    a.x += ra.x, a.y += ra.y, a.z += ra.z, a.w += ra.w;
                    b.x += rb.x, b.y += rb.y, b.z += rb.z, b.w += rb.w;
                }
                *reinterpret_cast<float4 *>(pg + (i * 64 + lane) * 4) = a;
                *reinterpret_cast<float4 *>(pb + (i * 64 + lane) * 4) = b;
            }
    }
}

// fp32 LayerNorm of a few rows (ln_post on the class rows: the projection gradient needs its fp32 output)
__global__ __launch_bounds__(256) void ln_f32_kernel(const float *x, long ldx, const float *gamma, const float *beta,
                                                     int rows, int width, float eps, float *out, long ldo)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = units(width, lane);
    float4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; i++)
        if (i < nv) {
            v[i] = *reinterpret_cast<const float4 *>(x + row * ldx + (i * 64 + lane) * 4);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    const float mean = wave_sum(s) / (float)width;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; i++)
        if (i < nv) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    const float rstd = 1.f / __builtin_sqrtf(wave_sum(q) / (float)width + eps);
#pragma unroll
    for (int i = 0; i < LN_MAXV; i++)
        if (i < nv) {
            const int c = (i * 64 + lane) * 4;
            const float4 g = *reinterpret_cast<const float4 *>(gamma + c);
            const float4 b = *reinterpret_cast<const float4 *>(beta + c);
            *reinterpret_cast<float4 *>(out + row * ldo + c) =
                make_float4((v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y,
                            (v[i].z - mean) * rstd * g.z + b.z, (v[i].w - mean) * rstd * g.w + b.w);
        }
}

// column sums of a [M, N] matrix over row slabs: partial[slab][n] (bias gradients).  A workgroup owns 64 x 16
// bytes of the row (512 16-bit or 256 fp32 columns: every wave load is 1 KiB contiguous) and a slab of rows; its
// four waves take every fourth row, eight rows in flight per lane, and meet in LDS.  Fixed order: bit-reproducible.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T *src, long ld, int M, int N, int slab_rows, float *partial)
{
    constexpr int V = 16 / sizeof(T);
    typedef T vec __attribute__((ext_vector_type(V)));
    __shared__ float red[3][64][V + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = (blockIdx.x * 64 + lane) * V;
    const bool live = n < N;
    const int r0 = blockIdx.y * slab_rows, r1 = r0 + slab_rows < M ? r0 + slab_rows : M;
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; j++) acc[j] = 0.f;
    if (live) {
        int r = r0 + wave;
        for (; r + 28 < r1; r += 32) {
            vec v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const vec *>(src + (long)(r + 4 * u) * ld + n);
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int j = 0; j < V; j++) acc[j] += (float)v[u][j];
        }
        for (; r < r1; r += 4) {
            const vec v = *reinterpret_cast<const vec *>(src + (long)r * ld + n);
#pragma unroll
            for (int j = 0; j < V; j++) acc[j] += (float)v[j];
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < V; j++) red[wave - 1][lane][j] = acc[j];
    }
    __syncthreads();
    if (wave == 0 && live) {
#pragma unroll
        for (int j = 0; j < V; j++)
            partial[(long)blockIdx.y * N + n + j] = (acc[j] + red[0][lane][j]) + (red[1][lane][j] + red[2][lane][j]);
    }
}

// out[i] = sum_p part[p * stride + i], p ascending (n, stride multiples of 4)
__global__ __launch_bounds__(256) void reduce_kernel(const float *part, long stride, int P, long n, float *out)
{
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int p = 0;
        for (; p + 4 <= P; p += 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const float4 *>(part + (p + u) * stride + i);
#pragma unroll
            for (int u = 0; u < 4; u++) s.x += v[u].x, s.y += v[u].y, s.z += v[u].z, s.w += v[u].w;
        }
        for (; p < P; p++) {
            const float4 v = *reinterpret_cast<const float4 *>(part + p * stride + i);
            s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
        }
        *reinterpret_cast<float4 *>(out + i) = s;
    }
}
// The same sum when there are MANY partial rows and few columns (LayerNorm's per-workgroup partials, the bias
// gradients' row slabs): a workgroup owns 64 columns, its 16 thread rows walk the partials 16 apart, LDS adds them up
// in a fixed order.
__global__ __launch_bounds__(256) void reduce_tall_kernel(const float *part, long stride, int P, long n, float *out)
{
    __shared__ float4 red[16][16];
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const long i = ((long)blockIdx.x * 16 + cg) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
        int p = rl;
        for (; p + 48 < P; p += 64) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const float4 *>(part + (long)(p + 16 * u) * stride + i);
#pragma unroll
            for (int u = 0; u < 4; u++) s.x += v[u].x, s.y += v[u].y, s.z += v[u].z, s.w += v[u].w;
        }
        for (; p < P; p += 16) {
            const float4 v = *reinterpret_cast<const float4 *>(part + (long)p * stride + i);
            s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
        }
    }
    red[rl][cg] = s;
    __syncthreads();
    if (rl == 0 && i < n) {
#pragma unroll
        for (int r = 1; r < 16; r++) {
            const float4 v = red[r][cg];
            s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
        }
        *reinterpret_cast<float4 *>(out + i) = s;
    }
}

// conv1: the patch row is [hi | lo | 0], so d conv1.weight[w][c] = sum_p (part[w][c] + part[w][k + c])
__global__ __launch_bounds__(256) void reduce_conv_kernel(const float *part, long stride, int P, int W, int k, int kpad,
                                                          float *out)
{
    const long n = (long)W * k;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long w = i / k, c = i - w * k;
        float s = 0.f;
        for (int p = 0; p < P; p++) s += part[p * stride + w * kpad + c] + part[p * stride + w * kpad + k + c];
        out[i] = s;
    }
}

// d positional_embedding[s] = sum over images of de[n, s]; d class_embedding = the s = 0 row
__global__ __launch_bounds__(256) void pos_grad_kernel(const float *de, int n_img, int S, int W, float *dpos, float *dcls)
{
    const int s = blockIdx.x;
    for (int c = threadIdx.x * 4; c < W; c += 1024) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int n = 0; n < n_img; n++) {
            const float4 v = *reinterpret_cast<const float4 *>(de + ((long)n * S + s) * W + c);
            a.x += v.x, a.y += v.y, a.z += v.z, a.w += v.w;
        }
        if (dpos) *reinterpret_cast<float4 *>(dpos + (long)s * W + c) = a;
        if (dcls && s == 0) *reinterpret_cast<float4 *>(dcls + c) = a;
    }
}

// fp32 [rows, cols] -> hi (rounded), lo (rounded remainder), hi transposed; blockIdx.z = matrix of a same-shape list
template <int DT>
__global__ __launch_bounds__(256) void pack_weight_kernel(const ec_pack_item *items, int rows, int cols)
{
    typedef typename T16<DT>::elem elem;
    const ec_pack_item it = items[blockIdx.z];
    const float *w = it.w;
    void *hi = it.hi, *lo = it.lo, *hi_t = it.hi_t;
    __shared__ elem tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int rl = i >> 6, cl = i & 63;
        const int r = r0 + rl, c = c0 + cl;
        elem h = (elem)0.f;
        if (r < rows && c < cols) {
            const float v = w[(long)r * cols + c];
            h = to16(v, elem());
            if (hi) ((elem *)hi)[(long)r * cols + c] = h;
            if (lo) ((elem *)lo)[(long)r * cols + c] = to16(v - (float)h, elem());
        }
        tile[rl][cl] = h;
    }
    if (!hi_t) return;
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int cl = i >> 6, rl = i & 63;
        const int r = r0 + rl, c = c0 + cl;
        if (r < rows && c < cols) ((elem *)hi_t)[(long)c * rows + r] = tile[rl][cl];
    }
}

// ------------------------------------------------------------------------------------------
// LoRA (models/lora.py): the factors act through merged weights W + up @ down and receive their gradients
// from the merged weight's: d up = dW down^T, d down = up^T dW.  r <= 64; fp32; a few MB per matrix.
// ------------------------------------------------------------------------------------------
constexpr int LORA_MAXR = 64;

// d_down of item z = sum over its slabs of partial[z][slab][r * cols]
__global__ __launch_bounds__(256) void lora_ddown_reduce_kernel(const ec_lora_item *items, const float *scratch, int slabs,
                                                                long n)
{
    const float *part = scratch + (long)blockIdx.y * slabs * n;
    float *out = items[blockIdx.y].d_down;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int p = 0; p < slabs; p++) {
            const float4 v = *reinterpret_cast<const float4 *>(part + p * n + i);
            s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
        }
        *reinterpret_cast<float4 *>(out + i) = s;
    }
}

// torch.optim.Adam over a list of tensors in one launch: blockIdx.y = tensor
__global__ __launch_bounds__(256) void adam_multi_kernel(const ec_adam_item *items, float lr0, float lr1, float b1, float b2,
                                                         float eps, float wd, float bc1, float bc2_sqrt, const int *skip,
                                                         const float *scalars)
{
    if (skip && *skip) return;     // a non-finite gradient somewhere: the whole step is dropped (GradScaler.step)
    if (scalars) lr0 = scalars[EC_STEP_LR0], lr1 = scalars[EC_STEP_LR1], bc1 = scalars[EC_STEP_BC1], bc2_sqrt = scalars[EC_STEP_BC2_SQRT];
    const ec_adam_item it = items[blockIdx.y];
    const float lr = it.group ? lr1 : lr0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < it.n; i += (long)gridDim.x * 256) {
        float gi = it.grad[i];
        if (wd != 0.f) gi += wd * it.param[i];
        const float mi = it.exp_avg[i] * b1 + (1.f - b1) * gi;
        const float vi = it.exp_avg_sq[i] * b2 + (1.f - b2) * gi * gi;
        it.exp_avg[i] = mi, it.exp_avg_sq[i] = vi;
        it.param[i] -= lr / bc1 * mi / (__builtin_sqrtf(vi) / bc2_sqrt + eps);
    }
}

// out[i][j] = base[i][j] + sum_k up[i][k] down[k][j]; a thread owns four columns of 16 rows, 16 factors at a time
__global__ __launch_bounds__(256) void lora_merge_kernel(const ec_lora_item *items, int rows, int cols, int r)
{
    const ec_lora_item it = items[blockIdx.z];
    const float *base = it.base, *up = it.up, *down = it.down;
    float *out = it.out;
    const int j = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (j >= cols) return;
    const int i0 = blockIdx.y * 16;
    float4 a[16];
#pragma unroll
    for (int u = 0; u < 16; u++)
        a[u] = i0 + u < rows ? *reinterpret_cast<const float4 *>(base + (long)(i0 + u) * cols + j)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k0 = 0; k0 < r; k0 += 16) {
        float4 d[16];
#pragma unroll
        for (int k = 0; k < 16; k++)
            d[k] = k0 + k < r ? *reinterpret_cast<const float4 *>(down + (long)(k0 + k) * cols + j)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (i0 + u >= rows) continue;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (k0 + k >= r) continue;
                const float w = up[(long)(i0 + u) * r + k0 + k];      // uniform over the workgroup
                a[u].x = __builtin_fmaf(w, d[k].x, a[u].x), a[u].y = __builtin_fmaf(w, d[k].y, a[u].y);
                a[u].z = __builtin_fmaf(w, d[k].z, a[u].z), a[u].w = __builtin_fmaf(w, d[k].w, a[u].w);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 16; u++)
        if (i0 + u < rows) *reinterpret_cast<float4 *>(out + (long)(i0 + u) * cols + j) = a[u];
}

// d_up[i][k] = sum_j dW[i][j] down[k][j]: a wave takes four rows (each `down` value it loads serves all four),
// eight factors at a time
__global__ __launch_bounds__(256) void lora_dup_kernel(const ec_lora_item *items, int rows, int cols, int r)
{
    const ec_lora_item it = items[blockIdx.z];
    const float *dW = it.dW, *down = it.down;
    float *d_up = it.d_up;
    const int lane = threadIdx.x & 63;
    const int i0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    if (i0 >= rows) return;
    for (int k0 = 0; k0 < r; k0 += 8) {
        float acc[4][8];
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int k = 0; k < 8; k++) acc[u][k] = 0.f;
        for (int j = lane * 4; j < cols; j += 256) {
            float4 w[4];
#pragma unroll
            for (int u = 0; u < 4; u++)
                w[u] = i0 + u < rows ? *reinterpret_cast<const float4 *>(dW + (long)(i0 + u) * cols + j)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (k0 + k >= r) continue;
                const float4 d = *reinterpret_cast<const float4 *>(down + (long)(k0 + k) * cols + j);
#pragma unroll
                for (int u = 0; u < 4; u++)
                    acc[u][k] += (w[u].x * d.x + w[u].y * d.y) + (w[u].z * d.z + w[u].w * d.w);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float t = wave_sum(acc[u][k]);
                if (lane == 0 && k0 + k < r && i0 + u < rows) d_up[(long)(i0 + u) * r + k0 + k] = t;
            }
    }
}

// partial[slab][k][j] = sum over the slab's rows i of up[i][k] dW[i][j]; a thread owns four columns, 16 factors at
// a time
__global__ __launch_bounds__(256) void lora_ddown_kernel(const ec_lora_item *items, int rows, int cols, int r,
                                                         int slab_rows, float *scratch)
{
    const ec_lora_item it = items[blockIdx.z];
    const float *dW = it.dW, *up = it.up;
    float *partial = scratch + (long)blockIdx.z * gridDim.y * r * cols;
    const int j = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (j >= cols) return;
    const int i0 = blockIdx.y * slab_rows, i1 = i0 + slab_rows < rows ? i0 + slab_rows : rows;
    for (int k0 = 0; k0 < r; k0 += 16) {
        float4 acc[16];
#pragma unroll
        for (int k = 0; k < 16; k++) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = i0; i < i1; i++) {
            const float4 w = *reinterpret_cast<const float4 *>(dW + (long)i * cols + j);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (k0 + k >= r) continue;
                const float u = up[(long)i * r + k0 + k];             // uniform over the workgroup
                acc[k].x = __builtin_fmaf(u, w.x, acc[k].x), acc[k].y = __builtin_fmaf(u, w.y, acc[k].y);
                acc[k].z = __builtin_fmaf(u, w.z, acc[k].z), acc[k].w = __builtin_fmaf(u, w.w, acc[k].w);
            }
        }
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (k0 + k < r)
                *reinterpret_cast<float4 *>(partial + ((long)blockIdx.y * r + k0 + k) * cols + j) = acc[k];
    }
}

// ------------------------------------------------------------------------------------------
// LoRA gradients straight from the activations.  y = x (W + up down)^T, so with P = x down^T and Q = dy up
// (both [rows, r]):  d up = dy^T P,  d down = Q^T x -- the W x W gradient of the merged weight never exists.
//   lowrank_project_kernel     coefficients c[m][0 .. 16 RT) = In[m][:] . F[.][:]^T  (MFMA: the 16 factors of a
//                              tile are one 16 x 16 x 32 row block; a wave owns 16 rows of In, read straight from
//                              HBM), for up to three factor sets that multiply the same In (q, k, v all read
//                              ln_1(x)).  Written TRANSPOSED, [16 RT][Mp] in the 16-bit compute type, as two
//                              planes: hi = c rounded, lo = (c - hi) * 2^LO_SHIFT (same exponent range as hi, so
//                              the pair carries ~22 bits even where f16 goes subnormal); rows m >= M are zero.
//   lowrank_outer_mfma_kernel  partial[slab][k][c] = sum over the slab's rows m of c[m][k] In[m][c], the row index
//                              being the MFMA's contraction index: the coefficient planes are the A operand as
//                              they lie; In's [32 rows][64 columns] tile goes through a wave-private LDS image and
//                              comes back column-major through ds_read_b64_tr_b16.  hi and lo accumulate apart
//                              and meet at the store.  Up to three coefficient sets share one In (d down of
//                              q, k, v).
// ------------------------------------------------------------------------------------------
template <int DT> struct LoShift;
template <> struct LoShift<0> { static constexpr float up = 2048.f, down = 1.f / 2048.f; };
template <> struct LoShift<1> { static constexpr float up = 256.f, down = 1.f / 256.f; };

struct ProjectItem {
    const void *in;     // [M, C] 16-bit at row stride ld
    long ld;
    int nf;             // factor sets applied to this input (1 .. 3; more than 1 only with RT == 1)
    const void *ff[3];  // the factor rows [16 RT, C] (zero rows past r) in fragment order (lowrank_frag_kernel)
    void *out_t[3];     // [2][16 RT][Mp] 16-bit: hi plane, lo plane
    int C;
};
struct ProjectArgs {
    ProjectItem it[4];
    int M, Mp;
};

// NF factor sets of RT 16-row tiles each (more than one set only with RT == 1).  The count is a template
// parameter of the body, not a runtime guard: with guards every MFMA sits behind a branch and the compiler can
// no longer count the loads in flight (it waits for all of them: measured, no overlap between rounds).
template <int DT, int NF, int RT>
__device__ __forceinline__ void lowrank_project_body(const ProjectArgs &a, const ProjectItem &it, unsigned char *lds)
{
    typedef typename T16<DT>::elem elem;
    typedef typename T16<DT>::v8 v8;
    static_assert(NF == 1 || RT == 1, "several factor sets only with one factor tile each");
    constexpr int NA = NF * RT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c16 = lane & 15;
    const int m0 = (blockIdx.x * 4 + wave) * 16;
    if (m0 >= a.Mp) return;
    unsigned char *img = lds + wave * 8192;
    // In's round [16 rows][128 columns] arrives row-contiguous (a quarter wave reads 256 contiguous bytes; the
    // operand order -- lane = row -- straight from memory costs one cache-line look-up per lane: measured, the
    // kernel ran at the texture unit's rate, not the memory's) and is re-read from a wave-private LDS image.
    // The factors come in fragment order (lowrank_frag_kernel): 1 KiB contiguous per load.
    const elem *src[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int m = m0 + g + 4 * i;
        m = m < a.M ? m : a.M - 1;
        src[i] = (const elem *)it.in + (long)m * it.ld + c16 * 8;
    }
    const int ksteps = it.C / 32;
    const v8 *wf[NA];
#pragma unroll
    for (int t = 0; t < NA; t++)
        wf[t] = (const v8 *)(NF == 1 ? it.ff[0] : it.ff[t]) + (long)(NF == 1 ? t : 0) * ksteps * 64 + lane;
    f32x4 acc[NA];
#pragma unroll
    for (int t = 0; t < NA; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto load = [&](int r, v8(&x)[4], v8(&w)[NA][4]) {
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] = *reinterpret_cast<const v8 *>(src[i] + r * 128);
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int t = 0; t < NA; t++) w[t][u] = wf[t][(long)(4 * r + u) * 64];
    };
    auto compute = [&](const v8(&x)[4], const v8(&w)[NA][4], unsigned char *im) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = g + 4 * i;
            *reinterpret_cast<v8 *>(im + row * 256 + ((c16 ^ row) << 4)) = x[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const v8 xf = *reinterpret_cast<const v8 *>(im + c16 * 256 + (((4 * u + g) ^ c16) << 4));
#pragma unroll
            for (int t = 0; t < NA; t++) acc[t] = mfma16(w[t][u], xf, acc[t]);
        }
    };
    // two rounds in flight: a wave lives for C / 128 rounds only, so a latency it waits out in full shows
    const int rounds = it.C / 128;
    v8 xa[4], xb[4], wa[NA][4], wb[NA][4];
    if (rounds > 0) load(0, xa, wa);
    if (rounds > 1) load(1, xb, wb);
    for (int r = 0; r < rounds; r += 2) {
        compute(xa, wa, img);
        if (r + 2 < rounds) load(r + 2, xa, wa);
        if (r + 1 < rounds) {
            compute(xb, wb, img + 4096);
            if (r + 3 < rounds) load(r + 3, xb, wb);
        }
    }
    // a last half round (C = 64 (2 n + 1)): operand order straight from memory
    if (rounds * 128 < it.C) {
        const int m = m0 + c16 < a.M ? m0 + c16 : a.M - 1;
        const elem *in = (const elem *)it.in + (long)m * it.ld + g * 8;
        for (int ks = rounds * 4; ks < ksteps; ks++) {
            const v8 x = *reinterpret_cast<const v8 *>(in + ks * 32);
#pragma unroll
            for (int t = 0; t < NA; t++) acc[t] = mfma16(wf[t][(long)ks * 64], x, acc[t]);
        }
    }
    // acc[t][r] = c[m0 + c16][4 g + r] of tile t; sixteen lanes write sixteen consecutive m
    const bool inside = m0 + c16 < a.M;
    const long plane = (long)16 * RT * a.Mp;
#pragma unroll
    for (int t = 0; t < NA; t++) {
        elem *dst = (elem *)(NF == 1 ? it.out_t[0] : it.out_t[t]) + (long)((NF == 1 ? 16 * t : 0) + 4 * g) * a.Mp + m0 + c16;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float v = inside ? acc[t][r] : 0.f;
            const elem hi = (elem)v;
            dst[(long)r * a.Mp] = hi;
            dst[plane + (long)r * a.Mp] = (elem)((v - (float)hi) * LoShift<DT>::up);
        }
    }
}

// MAXF = 3: items carry one to three factor sets (RT == 1); MAXF = 1: one set of RT tiles
template <int DT, int MAXF, int RT>
__global__ __launch_bounds__(256) void lowrank_project_kernel(const ProjectArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * 2 * 4096];
    const ProjectItem &it = a.it[blockIdx.y];
    if constexpr (MAXF == 1) {
        lowrank_project_body<DT, 1, RT>(a, it, lds);
    } else {
        if (it.nf == 3) lowrank_project_body<DT, 3, 1>(a, it, lds);
        else if (it.nf == 2) lowrank_project_body<DT, 2, 1>(a, it, lds);
        else lowrank_project_body<DT, 1, 1>(a, it, lds);
    }
}

// Factor rows [16 RT][C] -> MFMA fragment order: 16-byte piece (tile t, k step ks, lane) = rows 16 t + (lane & 15),
// columns 32 ks + 8 (lane >> 4) .. + 8, so that a wave's operand load is 1 KiB contiguous.
struct FragArgs {
    const void *src[8];
    void *dst[8];
    int C, RT;
};
__global__ __launch_bounds__(256) void lowrank_frag_kernel(const FragArgs a)
{
    const int ksteps = a.C / 32;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)a.RT * ksteps * 64) return;
    const int lane = (int)(idx & 63);
    const long fs = idx >> 6;
    const int t = (int)(fs / ksteps), ks = (int)(fs - (long)t * ksteps);
    const unsigned short *src = static_cast<const unsigned short *>(a.src[blockIdx.y]);
    reinterpret_cast<uint4 *>(a.dst[blockIdx.y])[idx] =
        *reinterpret_cast<const uint4 *>(src + (long)(16 * t + (lane & 15)) * a.C + 32 * ks + 8 * (lane >> 4));
}

struct OuterItem {
    const void *in;        // [M, C] 16-bit at row stride ld
    long ld;
    int ns;                // coefficient sets multiplying this input (1 .. 3; more than 1 only with RT == 1)
    const void *coef_t[3]; // [2][16 RT][Mp] 16-bit planes of lowrank_project_kernel
    float *partial[3];     // [slabs][r][C]
};
struct OuterArgs {
    OuterItem it[4];
    int M, Mp, C, r, slab_rows;       // slab_rows: a multiple of 32
};

// byte offset of 16-byte chunk ch (0 .. 7) of row (0 .. 31) in a wave's [32][64 x 16-bit] image.  A 32-lane half of
// a transposed read takes 32 bytes of rows {0..3, 8..11} + 4 hh (+ 16): the XOR spreads them over the 8 32-byte
// windows of the 64 banks.
__device__ __forceinline__ int outer_off(int row, int ch)
{
    return row * 128 + ((ch ^ ((((row >> 1) & 1) + 2 * ((row >> 3) & 1)) << 1)) << 4);
}

template <int DT, int NS, int RT>
__device__ __forceinline__ void lowrank_outer_body(const OuterArgs &a, const OuterItem &it, unsigned char *lds)
{
    typedef typename T16<DT>::elem elem;
    typedef typename T16<DT>::v8 v8;
    typedef typename T16<DT>::v4 v4;
    static_assert(NS == 1 || RT == 1, "several coefficient sets only with one factor tile each");
    constexpr int NA = NS * RT;                   // accumulator groups: RT tiles of one set, or one tile of NS sets
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c16 = lane & 15;
    const int c0 = blockIdx.x * 256 + wave * 64;
    if (c0 >= a.C) return;                        // a whole wave: EXEC stays full for the transposed reads
    unsigned char *buf = lds + wave * 8192;
    const int m_lo = blockIdx.y * a.slab_rows;
    const int m_hi = m_lo + a.slab_rows < a.Mp ? m_lo + a.slab_rows : a.Mp;
    const int lr = lane >> 3, lch = lane & 7;
    const elem *ct[NA];
    const long plane = (long)16 * RT * a.Mp;
#pragma unroll
    for (int t = 0; t < NA; t++) {
        const elem *base = (const elem *)(NS == 1 ? it.coef_t[0] : it.coef_t[t]);
        ct[t] = base + (long)((NS == 1 ? 16 * t : 0) + c16) * a.Mp + 8 * g;
    }
    auto fetch = [&](int m0, v8(&x)[4]) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int row = m0 + lr + 8 * i;
            row = row < a.M ? row : a.M - 1;      // (its coefficients are zero)
            x[i] = *reinterpret_cast<const v8 *>((const elem *)it.in + (long)row * it.ld + c0 + lch * 8);
        }
    };
    f32x4 acc[2][NA][4];
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int t = 0; t < NA; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[h][t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // one 32-row step: x (fetched two steps ago) into the image, the fetch two steps ahead into x, the products
    auto step = [&](int m0, v8(&x)[4], unsigned char *img) {
#pragma unroll
        for (int i = 0; i < 4; i++) *reinterpret_cast<v8 *>(img + outer_off(lr + 8 * i, lch)) = x[i];
        if (m0 + 64 < m_hi) fetch(m0 + 64, x);
        v8 ch[NA], cl[NA];
#pragma unroll
        for (int t = 0; t < NA; t++)
        {
            ch[t] = *reinterpret_cast<const v8 *>(ct[t] + m0);
            cl[t] = *reinterpret_cast<const v8 *>(ct[t] + plane + m0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < 4; j++) {
            v8 f;
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const int row = 8 * g + 4 * hh + (c16 >> 2);
                const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(
                    img + outer_off(row, 2 * j + ((c16 & 3) >> 1)) + 8 * (c16 & 1)));
                const v4 tv = __builtin_bit_cast(v4, t);
                f[4 * hh] = tv[0], f[4 * hh + 1] = tv[1], f[4 * hh + 2] = tv[2], f[4 * hh + 3] = tv[3];
            }
#pragma unroll
            for (int t = 0; t < NA; t++)
            {
                acc[0][t][j] = mfma16(ch[t], f, acc[0][t][j]);
                acc[1][t][j] = mfma16(cl[t], f, acc[1][t][j]);
            }
        }
    };
    v8 xa[4], xb[4];
    fetch(m_lo, xa);
    if (m_lo + 32 < m_hi) fetch(m_lo + 32, xb);
    for (int m0 = m_lo; m0 < m_hi; m0 += 64) {
        step(m0, xa, buf);
        if (m0 + 32 < m_hi) step(m0 + 32, xb, buf + 4096);
    }
    // acc[.][t][j][q] = T[k = 4 g + q of tile t][c0 + 16 j + c16]
#pragma unroll
    for (int t = 0; t < NA; t++) {
        float *part = (NS == 1 ? it.partial[0] : it.partial[t]) + (long)blockIdx.y * a.r * a.C;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int k = (NS == 1 ? 16 * t : 0) + 4 * g + q;
            if (k >= a.r) continue;
#pragma unroll
            for (int j = 0; j < 4; j++)
                part[(long)k * a.C + c0 + 16 * j + c16] = acc[0][t][j][q] + acc[1][t][j][q] * LoShift<DT>::down;
        }
    }
}

// MAXS = 3: items carry one to three coefficient sets (RT == 1); MAXS = 1: one set of RT tiles
template <int DT, int MAXS, int RT>
__global__ __launch_bounds__(256, 2) void lowrank_outer_mfma_kernel(const OuterArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * 2 * 4096];
    const OuterItem &it = a.it[blockIdx.z];
    if constexpr (MAXS == 1) {
        lowrank_outer_body<DT, 1, RT>(a, it, lds);
    } else {
        if (it.ns == 3) lowrank_outer_body<DT, 3, 1>(a, it, lds);
        else if (it.ns == 2) lowrank_outer_body<DT, 2, 1>(a, it, lds);
        else lowrank_outer_body<DT, 1, 1>(a, it, lds);
    }
}

// out = sum over slabs of partial[slab][r][C], written [r, C] (transposed = 0) or [C, r] (transposed = 1)
struct OuterReduceArgs {
    const float *partial[4];
    float *out[4];
    int transposed[4];
    int slabs, r, C;
};
__global__ __launch_bounds__(256) void lowrank_reduce_kernel(const OuterReduceArgs a)
{
    const float *part = a.partial[blockIdx.y];
    float *out = a.out[blockIdx.y];
    const long n = (long)a.r * a.C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float s = 0.f;
        int p = 0;
        for (; p + 8 <= a.slabs; p += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = part[(p + u) * n + i];
#pragma unroll
            for (int u = 0; u < 8; u++) s += v[u];
        }
        for (; p < a.slabs; p++) s += part[p * n + i];
        const long k = i / a.C, c = i - k * a.C;
        out[a.transposed[blockIdx.y] ? c * a.r + k : i] = s;
    }
}

__global__ __launch_bounds__(256) void unscale_check_kernel(float *g, long n, float inv_scale, int *found_inf,
                                                            const float *scalars)
{
    if (scalars) inv_scale = scalars[EC_STEP_INV_SCALE];
    bool bad = false;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = g[i] * inv_scale;
        g[i] = v;
        bad |= !__builtin_isfinite(v);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(found_inf, 1);
}

// through F.normalize + the validity mask (clip_cls_ft.py:214-217): d f = valid ? (dfn - fn (fn . dfn)) / |f| : 0
// (row_idx: feats / dfeats hold the valid views only, view r lives in row row_idx[r]; invalid views have no row)
__global__ __launch_bounds__(256) void feat_grad_kernel(const float *feats, const float *fn, const float *dfn,
                                                        const unsigned char *valid, const int *row_idx, int R, int D,
                                                        float scale, float *dfeats, const float *scalars)
{
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= R) return;
    if (scalars) scale = scalars[EC_STEP_GRAD_SCALE];
    if (row_idx && !valid[r]) return;
    const long row = row_idx ? row_idx[r] : r;
    float s = 0.f, dot = 0.f;
    for (int j = lane; j < D; j += 64) {
        const float m = feats[row * D + j];
        s += m * m, dot += fn[(long)r * D + j] * dfn[(long)r * D + j];
    }
    const float inv = valid[r] ? scale / fmaxf(__builtin_sqrtf(wave_sum(s)), 1e-12f) : 0.f;
    dot = wave_sum(dot);
    for (int j = lane; j < D; j += 64)
        dfeats[row * D + j] = (dfn[(long)r * D + j] - fn[(long)r * D + j] * dot) * inv;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
inline int padded_rows(int M)
{
    const int mp = (M + 63) / 64 * 64;
    return M >= 2048 ? (M + 1023) / 1024 * 1024 : mp;
}
// K-batches for a weight-gradient GEMM with an [n_out, n_in] output over mp reduction rows: enough
// that the launch has about one tile per CU, each batch at least 256 rows long
inline int pick_splits(int n_out, int n_in, int mp, int cus)
{
    const int tiles = ((n_out + 255) / 256) * ((n_in + 255) / 256);
    int s = 1;
    while (s < 16 && tiles * s * 2 <= cus && (mp / 64) % (s * 2) == 0 && mp / (s * 2) >= 256) s *= 2;
    return s;
}

// the same for the row batches of a weight gradient taken straight from the row-major operands
inline int pick_splits_rows(int n_out, int n_in, int rows, int cus)
{
    // any count (the row batches may be ragged: rows past the end read as zero): as many batches as fill one round
    // of workgroups -- 48 tiles of a 3W x W gradient run as 5 batches on 240 of 256 CUs, not 4 on 192
    const int tiles = ((n_out + 255) / 256) * ((n_in + 255) / 256);
    int s = cus / tiles;
    s = s > 16 ? 16 : s;
    while (s > 1 && rows / s < 256) s--;
    return s < 1 ? 1 : s;
}

struct TrainBufs {
    // tape
    float *pre;                 // [M, W] un-normalised embedding
    float *x[65];               // x[l]: input of block l; x[L]: the tower's last residual stream
    float *xm[64];              // after the attention branch
    void *qkv[64], *att[64], *u[64];
    void *gact[64];             // QuickGELU(u), 16 bit: c_proj's input, kept for its weight gradient
    void *h1[64], *h2[64];      // the two LayerNorm outputs (16 bit): the weight / LoRA gradients' right-hand operands
    float *lse[64];
    // scratch shared by both passes
    void *g16;                  // QuickGELU output [M, 4W] (forward) / du, dqkv (backward)
    void *cls_hi, *cls_lo;
    // backward scratch
    float *dx, *dh32, *delta, *clsln, *dclsln, *part, *lnpart, *colpart;
    void *dx16, *da16, *ta, *tb;
    int Mp, ln_wgs, col_slabs, W;
    size_t part_floats;
    // LoRA gradients from the activations: four coefficient slots ([2][<= 64][Mp] 16-bit, P then Q) and the outer
    // products' partial sums
    void *lr_proj, *lr_frag;
    float *lr_part;
    int lr_slab, lr_slabs;
};

size_t carve_train(Scratch &sc, const ec_vit_weights *w, int n, TrainBufs &b)
{
    const int g = w->image_size / w->patch, G = g * g, S = G + 1, W = w->width, L = w->layers;
    const size_t M = (size_t)n * S;
    const int Mp = padded_rows((int)M);
    b.Mp = Mp;
    b.pre = (float *)sc.take(M * W * 4);
    for (int l = 0; l <= L; l++) b.x[l] = (float *)sc.take(M * W * 4);
    for (int l = 0; l < L; l++) {
        b.xm[l] = (float *)sc.take(M * W * 4);
        b.qkv[l] = sc.take(M * 3 * W * 2);
        b.att[l] = sc.take(M * W * 2);
        b.u[l] = sc.take(M * 4 * W * 2);
        b.gact[l] = sc.take(M * 4 * W * 2);
        b.h1[l] = sc.take(M * W * 2);
        b.h2[l] = sc.take(M * W * 2);
        b.lse[l] = (float *)sc.take((size_t)n * w->heads * S * 4);
    }
    b.g16 = sc.take(M * 4 * W * 2);          // >= n G W 4 bytes: also the patch GEMM's fp32 output
    b.cls_hi = sc.take((size_t)n * W * 2);
    b.cls_lo = sc.take((size_t)n * W * 2);
    b.dx = (float *)sc.take(M * W * 4);
    b.dh32 = (float *)sc.take(M * W * 4);
    b.delta = (float *)sc.take((size_t)n * w->heads * S * 4);
    b.clsln = (float *)sc.take((size_t)n * W * 4);
    b.dclsln = (float *)sc.take((size_t)n * W * 4);
    b.dx16 = sc.take(M * W * 2);
    b.da16 = sc.take(M * W * 2);
    // transposed copies: only the patch embedding's weight gradient still takes them (its rows skip the class tokens)
    b.ta = sc.take((size_t)W * Mp * 2);
    b.tb = sc.take((size_t)w->kpad * Mp * 2);
    size_t pf = 0;
    const int shapes[5][2] = {{3 * W, W}, {W, W}, {4 * W, W}, {W, 4 * W}, {W, w->kpad}};
    for (int i = 0; i < 5; i++) {
        const size_t f = (size_t)pick_splits(shapes[i][0], shapes[i][1], Mp, SIZING_CUS) * shapes[i][0] * shapes[i][1];
        const size_t fr = (size_t)pick_splits_rows(shapes[i][0], shapes[i][1], (int)M, SIZING_CUS) * shapes[i][0] * shapes[i][1];
        pf = f > pf ? f : pf;
        pf = fr > pf ? fr : pf;
    }
    b.part_floats = pf;
    b.part = (float *)sc.take(pf * 4);
    b.ln_wgs = (int)((M + 3) / 4 < 1024 ? (M + 3) / 4 : 1024);
    b.lnpart = (float *)sc.take((size_t)b.ln_wgs * 2 * W * 4);
    b.W = W;
    b.col_slabs = 256;
    b.colpart = (float *)sc.take((size_t)b.col_slabs * 4 * W * 4);
    b.lr_slab = 256;                                                       // the shortest row slab of an outer product
    b.lr_slabs = (int)((M + b.lr_slab - 1) / b.lr_slab);
    b.lr_proj = sc.take((M + 32) * 64 * 2 * 2 * 4);                        // four [2][<= 64][Mp] 16-bit coefficient slots
    b.lr_part = (float *)sc.take((size_t)b.lr_slabs * 64 * W * 4 * 4);     // four [slabs, <= 64, W] partial blocks
    b.lr_frag = sc.take((size_t)8 * 64 * W * 2);                           // eight factors [<= 64, W] in fragment order
    return sc.off;
}

int check_geometry(const ec_vit_weights *w, const char *who)
{
    EC_REQUIRE(w && w->blocks, "%s: weights are null", who);
    EC_REQUIRE(w->image_size % w->patch == 0, "%s: image %d not a multiple of patch %d", who, w->image_size, w->patch);
    EC_REQUIRE(w->width == w->heads * 64 && w->width % 64 == 0 && w->width <= LN_MAXV * 256,
               "%s: width %d (heads %d): head dim must be 64", who, w->width, w->heads);
    EC_REQUIRE(w->layers >= 1 && w->layers <= 64, "%s: %d layers", who, w->layers);
    EC_REQUIRE(w->kpad % 64 == 0 && w->kpad >= 6 * w->patch * w->patch, "%s: bad kpad %d", who, w->kpad);
    EC_REQUIRE(w->conv_w && w->conv_w_lo && w->proj_w && w->proj_w_lo, "%s: conv / proj weights need hi and lo parts", who);
    EC_REQUIRE(w->out_dim % 16 == 0, "%s: out_dim %d", who, w->out_dim);
    EC_REQUIRE(!w->precise, "%s: the split-precision tower has no training form", who);
    EC_REQUIRE(!w->q_scaled, "%s: the training form differentiates a plain q (ec_vit_weights.q_scaled must be 0)", who);
    return EC_OK;
}

template <int MODE>
int transpose(int dtype, const void *src, long ld, int M, int N, int Mp, int seq_out, int seq_in, int seq_off,
              void *dst_t, void *dst_rm, hipStream_t s)
{
    const dim3 grid((unsigned)(Mp / 64), (unsigned)((N + 63) / 64));
    ec::ProfScope prof(ec::PROF_TRANSPOSE, s, 0, (double)M * N * (MODE == 2 ? 4.0 : 2.0) + 2.0 * Mp * N);
    if (dtype == EC_F16)
        hipLaunchKernelGGL((transpose_kernel<EC_F16, MODE>), grid, dim3(256), 0, s, src, ld, M, N, Mp, seq_out, seq_in,
                           seq_off, dst_t, dst_rm);
    else
        hipLaunchKernelGGL((transpose_kernel<EC_BF16, MODE>), grid, dim3(256), 0, s, src, ld, M, N, Mp, seq_out, seq_in,
                           seq_off, dst_t, dst_rm);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

int reduce(const float *part, long stride, int P, long n, float *out, hipStream_t s)
{
    if (n % 4 != 0 || stride % 4 != 0) return ec::fail(EC_ERR_INVALID, "reduce: %ld elements at stride %ld", n, stride);
    const long blocks = (n / 4 + 255) / 256;
    ec::ProfScope prof(ec::PROF_REDUCE, s, 0, 4.0 * n * (P + 1));
    if (P >= 64 && n <= 65536) {
        hipLaunchKernelGGL(reduce_tall_kernel, dim3((unsigned)((n / 4 + 15) / 16)), dim3(256), 0, s, part, stride, P, n, out);
        EC_CHECK_HIP(hipGetLastError());
        return EC_OK;
    }
    hipLaunchKernelGGL(reduce_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, s, part, stride, P, n,
                       out);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

// db[N] = column sums of a [M, N] 16-bit or fp32 matrix
template <typename T> int bias_grad(const T *src, long ld, int M, int N, const TrainBufs &b, float *out, hipStream_t s)
{
    // about 1024 workgroups (four per CU of the device this was sized for), within the partial buffer
    constexpr int V = 16 / sizeof(T);
    const int col_groups = (N / V + 63) / 64;
    long want = 1024 / col_groups;
    const long fit = (long)b.col_slabs * 4 * b.W / N;                 // slabs of N floats the buffer holds
    want = want < fit ? want : fit;
    want = want < 1 ? 1 : want;
    int slab = (int)((M + want - 1) / want);
    slab = slab < 8 ? 8 : slab;
    const int slabs = (M + slab - 1) / slab;
    {
        ec::ProfScope prof(ec::PROF_REDUCE, s, 0, (double)M * N * sizeof(T));
        hipLaunchKernelGGL(colsum_kernel<T>, dim3((unsigned)col_groups, (unsigned)slabs), dim3(256), 0, s, src, ld, M, N, slab,
                           b.colpart);
    }
    return reduce(b.colpart, N, slabs, N, out, s);
}

// dW[n_out, n_in] = A_t[n_out, Mp] . B_t[n_in, Mp]^T in K-batches, reduced into `out` (or, conv: folded)
int weight_grad(int dtype, const void *a_t, const void *b_t, int n_out, int n_in, const TrainBufs &b, float *out,
                int conv_k, ec_stream_t stream)
{
    const int cus = ec::cu_count();
    EC_REQUIRE(cus > 0, "ec_vit_train_backward: cannot read the device's compute-unit count");
    int splits = pick_splits(n_out, n_in, b.Mp, cus);
    while ((size_t)splits * n_out * n_in > b.part_floats) splits /= 2;   // a device with more CUs than sized for
    ec_gemm_args g = {};
    g.M = n_out, g.N = n_in, g.K = b.Mp / splits, g.dtype = dtype, g.epilogue = EC_EPI_STORE32, g.variant = 0;
    g.A = a_t, g.lda = b.Mp, g.W = b_t, g.ldw = b.Mp, g.C = b.part, g.ldc = n_in;
    g.splits = splits, g.split_stride = (long)n_out * n_in;
    EC_TRY(ec_gemm(&g, stream));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (conv_k) {
        const long n = (long)n_out * conv_k;
        hipLaunchKernelGGL(reduce_conv_kernel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)),
                           dim3(256), 0, s, b.part, (long)n_out * n_in, splits, n_out, conv_k, n_in, out);
        EC_CHECK_HIP(hipGetLastError());
        return EC_OK;
    }
    return reduce(b.part, (long)n_out * n_in, splits, (long)n_out * n_in, out, s);
}

// dW[n_out, n_in] = dy^T x straight from the row-major activations (ec_gemm_args.transposed: the reduction runs over
// the operands' rows, their tiles are read column-major out of LDS) -- no transposed copies.  K-batches over row
// ranges, reduced into `out`.
int weight_grad_rows(int dtype, const void *dy, long ldy, const void *x, long ldx, int n_out, int n_in, int rows,
                     const TrainBufs &b, float *out, ec_stream_t stream)
{
    const int cus = ec::cu_count();
    EC_REQUIRE(cus > 0, "ec_vit_train_backward: cannot read the device's compute-unit count");
    int splits = pick_splits_rows(n_out, n_in, rows, cus);
    while ((size_t)splits * n_out * n_in > b.part_floats) splits /= 2;   // a device with more CUs than sized for
    ec_gemm_args g = {};
    g.M = n_out, g.N = n_in, g.K = ((rows + splits - 1) / splits + 63) / 64 * 64, g.dtype = dtype;
    g.epilogue = EC_EPI_STORE32, g.variant = 0, g.transposed = 1, g.k_rows = rows;
    g.A = dy, g.lda = ldy, g.W = x, g.ldw = ldx, g.C = b.part, g.ldc = n_in;
    g.splits = splits, g.split_stride = (long)n_out * n_in;
    EC_TRY(ec_gemm(&g, stream));
    return reduce(b.part, (long)n_out * n_in, splits, (long)n_out * n_in, out, static_cast<hipStream_t>(stream));
}

// ---- LoRA gradients of one block (ec_block_lora): P / Q projections, outer products, reduction ----
template <int DT> void launch_project(const ProjectArgs &a, int n_items, bool shared, int RT, hipStream_t s)
{
    const dim3 grid((unsigned)((a.Mp + 63) / 64), (unsigned)n_items);
    if (shared) hipLaunchKernelGGL((lowrank_project_kernel<DT, 3, 1>), grid, dim3(256), 0, s, a);
    else if (RT == 1) hipLaunchKernelGGL((lowrank_project_kernel<DT, 1, 1>), grid, dim3(256), 0, s, a);
    else if (RT == 2) hipLaunchKernelGGL((lowrank_project_kernel<DT, 1, 2>), grid, dim3(256), 0, s, a);
    else if (RT == 3) hipLaunchKernelGGL((lowrank_project_kernel<DT, 1, 3>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((lowrank_project_kernel<DT, 1, 4>), grid, dim3(256), 0, s, a);
}
template <int DT> void launch_outer(const OuterArgs &a, int n_items, int slabs, bool shared, int RT, hipStream_t s)
{
    const dim3 grid((unsigned)((a.C + 255) / 256), (unsigned)slabs, (unsigned)n_items);
    if (shared) hipLaunchKernelGGL((lowrank_outer_mfma_kernel<DT, 3, 1>), grid, dim3(256), 0, s, a);
    else if (RT == 1) hipLaunchKernelGGL((lowrank_outer_mfma_kernel<DT, 1, 1>), grid, dim3(256), 0, s, a);
    else if (RT == 2) hipLaunchKernelGGL((lowrank_outer_mfma_kernel<DT, 1, 2>), grid, dim3(256), 0, s, a);
    else if (RT == 3) hipLaunchKernelGGL((lowrank_outer_mfma_kernel<DT, 1, 3>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((lowrank_outer_mfma_kernel<DT, 1, 4>), grid, dim3(256), 0, s, a);
}

// One group of projections sharing the row count: inputs x_p (what the projection multiplies) and dy_p (the
// gradient of its output), factors down16_p [16 RT, C] and up16t_p [16 RT, C].  d_up_p [C, r], d_down_p [r, C].
struct LoraJob {
    const void *x, *dy;
    long ldx, ldy;
    const void *down16, *up16t;
    float *d_up, *d_down;
};
int lora_grads(int dtype, const LoraJob *jobs, int n, int M, int W, int r, const TrainBufs &b, hipStream_t s)
{
    const int RT = (r + 15) / 16;
    const int Mp = (M + 31) / 32 * 32;
    EC_REQUIRE(W % 64 == 0, "LoRA gradients: width %d is not a multiple of 64", W);
    const int cus = ec::cu_count();
    EC_REQUIRE(cus > 0, "LoRA gradients: cannot read the device's compute-unit count");
    ec::ProfScope prof(ec::PROF_SGEMM, s, 8.0 * M * W * r * n, 4.0 * M * W * n);
    // half 0: P_p = x_p down_p^T, d up_p = dy_p^T P_p;   half 1: Q_p = dy_p up_p, d down_p = Q_p^T x_p.
    // With one 16-row factor tile per projection (r <= 16) the projections that multiply the same x (q, k, v)
    // read it once: one project item with three factor sets in half 0, one outer item with three coefficient
    // sets in half 1.
    const bool share = RT == 1;
    const size_t slot = (size_t)2 * 64 * Mp;                       // 16-bit elements per coefficient slot
    // every factor of the block in fragment order: slot 2 i = down of job i, 2 i + 1 = up^T
    const size_t fslot = (size_t)64 * W;                           // 16-bit elements
    {
        FragArgs fa = {};
        fa.C = W, fa.RT = RT;
        for (int i = 0; i < n; i++) {
            fa.src[2 * i] = jobs[i].down16, fa.src[2 * i + 1] = jobs[i].up16t;
            fa.dst[2 * i] = static_cast<unsigned short *>(b.lr_frag) + (size_t)(2 * i) * fslot;
            fa.dst[2 * i + 1] = static_cast<unsigned short *>(b.lr_frag) + (size_t)(2 * i + 1) * fslot;
        }
        const long pieces = (long)RT * (W / 32) * 64;
        hipLaunchKernelGGL(lowrank_frag_kernel, dim3((unsigned)((pieces + 255) / 256), (unsigned)(2 * n)), dim3(256), 0, s, fa);
    }
    for (int half = 0; half < 2; half++) {
        ProjectArgs pa = {};
        pa.M = M, pa.Mp = Mp;
        OuterArgs oa = {};
        oa.M = M, oa.Mp = Mp, oa.C = W, oa.r = r;
        OuterReduceArgs ra = {};
        ra.r = r, ra.C = W;
        int np = 0, no = 0;
        for (int i = 0; i < n; i++) {
            const LoraJob &j = jobs[i];
            void *coef = static_cast<unsigned short *>(b.lr_proj) + (size_t)i * slot;
            float *part = b.lr_part + (size_t)i * b.lr_slabs * 64 * W;
            ra.partial[i] = part;
            ra.out[i] = half ? j.d_down : j.d_up;
            ra.transposed[i] = half ? 0 : 1;
            // the projection reads x (half 0) or dy (half 1); the outer product reads the other one
            const void *pin = half ? j.dy : j.x, *oin = half ? j.x : j.dy;
            const long pld = half ? j.ldy : j.ldx, old = half ? j.ldx : j.ldy;
            int pi = np, oi = no;
            if (share) {
                for (int q = 0; q < np; q++)
                    if (pa.it[q].in == pin && pa.it[q].ld == pld && pa.it[q].nf < 3) pi = q;
                for (int q = 0; q < no; q++)
                    if (oa.it[q].in == oin && oa.it[q].ld == old && oa.it[q].ns < 3) oi = q;
            }
            ProjectItem &p = pa.it[pi];
            if (pi == np) np++, p.in = pin, p.ld = pld, p.C = W, p.nf = 0;
            p.ff[p.nf] = static_cast<unsigned short *>(b.lr_frag) + (size_t)(2 * i + half) * fslot, p.out_t[p.nf] = coef, p.nf++;
            OuterItem &o = oa.it[oi];
            if (oi == no) no++, o.in = oin, o.ld = old, o.ns = 0;
            o.coef_t[o.ns] = coef, o.partial[o.ns] = part, o.ns++;
        }
        bool share_p = false, share_o = false;
        for (int q = 0; q < np; q++) share_p |= pa.it[q].nf > 1;
        for (int q = 0; q < no; q++) share_o |= oa.it[q].ns > 1;
        // row slabs: as many workgroups as the CUs hold at once (two each with three coefficient sets in
        // registers, four otherwise) in ONE round -- a few more and a second round runs nearly empty; at least
        // b.lr_slab rows each (the partial sums' scratch is sized for that)
        const int col_groups = (W + 255) / 256;
        int want = (share_o || RT > 1 ? 2 : 4) * cus / (no * col_groups);
        want = want < 1 ? 1 : want;
        int slab = ((Mp + want - 1) / want + 31) / 32 * 32;
        slab = slab < b.lr_slab ? b.lr_slab : slab;
        const int slabs = (Mp + slab - 1) / slab;
        oa.slab_rows = slab, ra.slabs = slabs;
        if (dtype == EC_F16)
            launch_project<EC_F16>(pa, np, share_p, RT, s), launch_outer<EC_F16>(oa, no, slabs, share_o, RT, s);
        else
            launch_project<EC_BF16>(pa, np, share_p, RT, s), launch_outer<EC_BF16>(oa, no, slabs, share_o, RT, s);
        hipLaunchKernelGGL(lowrank_reduce_kernel, dim3((unsigned)(((long)r * W + 255) / 256), (unsigned)n), dim3(256), 0, s, ra);
    }
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

int ln_backward(const float *x, long ldx, const float *dy, long ldy, const float *gamma, int rows, int W, float *dx,
                long ldo, int accumulate, float *dg, float *db, float *partials, int max_wgs, hipStream_t s,
                void *dx16 = nullptr, int dtype = EC_F16)
{
    int wgs = (rows + 3) / 4;
    if (wgs > max_wgs) wgs = max_wgs;
    const bool want = dg || db;
    {
        ec::ProfScope prof(ec::PROF_LN_BWD, s, 0, (double)rows * W * (accumulate ? 16.0 : 12.0));
        if (dtype == EC_BF16)
            hipLaunchKernelGGL(ln_bwd_kernel<EC_BF16>, dim3((unsigned)wgs), dim3(256), 0, s, x, ldx, dy, ldy, gamma, rows, W,
                               LN_EPS, dx, ldo, accumulate, want ? partials : (float *)nullptr, dx16);
        else
            hipLaunchKernelGGL(ln_bwd_kernel<EC_F16>, dim3((unsigned)wgs), dim3(256), 0, s, x, ldx, dy, ldy, gamma, rows, W,
                               LN_EPS, dx, ldo, accumulate, want ? partials : (float *)nullptr, dx16);
    }
    EC_CHECK_HIP(hipGetLastError());
    // (d gamma | d beta lie side by side in a partial row; side by side in the gradient buffer too -> one launch)
    if (dg && db == dg + W) return reduce(partials, 2L * W, wgs, 2L * W, dg, s);
    if (dg) EC_TRY(reduce(partials, 2L * W, wgs, W, dg, s));
    if (db) EC_TRY(reduce(partials + W, 2L * W, wgs, W, db, s));
    return EC_OK;
}

int gemm_x(int M, int N, int K, int dtype, int epi, const void *A, const void *W, void *C, const float *resid, void *aux,
           const float *bias, ec_stream_t s)
{
    ec_gemm_args g = {};
    g.M = M, g.N = N, g.K = K, g.dtype = dtype, g.epilogue = epi, g.variant = 0;
    g.A = A, g.lda = K, g.W = W, g.ldw = K, g.bias = bias, g.C = C, g.ldc = N, g.resid = resid, g.aux = aux;
    return ec_gemm(&g, s);
}

// C[m0 + m][n] = (resid ? resid : 0) + bias[n] + sum over K-batches of partial[s][m][n]
__global__ __launch_bounds__(256) void tail_fixup_kernel(const float *partial, int splits, int rows, int N, const float *bias,
                                                         const float *resid, float *C)
{
    const long n4 = (long)rows * N / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int col = (int)((i * 4) % N);
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int p = 0; p < splits; p++) {
            const float4 v = *reinterpret_cast<const float4 *>(partial + (long)p * rows * N + i * 4);
            a.x += v.x, a.y += v.y, a.z += v.z, a.w += v.w;
        }
        if (bias) {
            const float4 b = *reinterpret_cast<const float4 *>(bias + col);
            a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
        }
        if (resid) {
            const float4 r = *reinterpret_cast<const float4 *>(resid + i * 4);
            a.x += r.x, a.y += r.y, a.z += r.z, a.w += r.w;
        }
        *reinterpret_cast<float4 *>(C + i * 4) = a;
    }
}

// fp32-output GEMM (STORE32 / RESID32) over a row count that leaves a short last row of tiles, K >= 2048.  The
// persistent kernel would spend a whole extra round on those few tiles, each streaming all of K alone (its K loop
// is latency bound: +64 us at K = 4096 whatever the row count).  Here the full row tiles go through one launch and the
// last `rem` rows through a second one cut into K-batches -- tiles_n x splits workgroups, each a fraction of K --
// whose fp32 partial sums a small kernel adds up with bias and residual.  (The rows of the tail see a different
// summation order than rows elsewhere in the batch: fine for training, not used on the inference path, whose results
// do not depend on where in a batch a frame sits.)
int gemm_rows32(int M, int N, int K, int dtype, int epi, const void *A, const void *W, float *C, const float *resid,
                const float *bias, float *partial, size_t partial_floats, ec_stream_t stream)
{
    const int cus = ec::cu_count();
    const int rem = M % 256, full = M - rem, tiles_n = (N + 255) / 256;
    const long tiles_main = (long)(full / 256) * tiles_n;
    const bool extra_round = cus > 0 && full > 0 && rem > 0 && rem <= 128 &&
                             (tiles_main + cus - 1) / cus < (tiles_main + tiles_n + cus - 1) / cus;
    int splits = 1;
    if (extra_round && K >= 2048)
        while (splits < 8 && tiles_n * splits * 2 <= cus && (K / 64) % (splits * 2) == 0 && K / (splits * 2) >= 256) splits *= 2;
    if (splits < 2 || (size_t)splits * rem * N > partial_floats || (epi != EC_EPI_STORE32 && epi != EC_EPI_RESID32))
        return gemm_x(M, N, K, dtype, epi, A, W, C, resid, nullptr, bias, stream);
    const size_t esz = 2;
    EC_TRY(gemm_x(full, N, K, dtype, epi, A, W, C, resid, nullptr, bias, stream));
    ec_gemm_args g = {};
    g.M = rem, g.N = N, g.K = K / splits, g.dtype = dtype, g.epilogue = EC_EPI_STORE32, g.variant = 0;
    g.A = static_cast<const unsigned char *>(A) + (size_t)full * K * esz, g.lda = K, g.W = W, g.ldw = K;
    g.C = partial, g.ldc = N, g.splits = splits, g.split_stride = (long)rem * N;
    EC_TRY(ec_gemm(&g, stream));
    const float *rsrc = epi == EC_EPI_RESID32 ? (resid ? resid : C) + (size_t)full * N : nullptr;
    const long n4 = (long)rem * N / 4;
    hipLaunchKernelGGL(tail_fixup_kernel, dim3((unsigned)((n4 + 255) / 256 < 1024 ? (n4 + 255) / 256 : 1024)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), partial, splits, rem, N, bias, rsrc, C + (size_t)full * N);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

}  // namespace

extern "C" {

EC_API size_t ec_vit_train_workspace_bytes(const ec_vit_weights *w, int n_img)
{
    if (!w || n_img <= 0 || w->patch <= 0 || w->layers < 1 || w->layers > 64) return 0;
    Scratch sc{nullptr, 0, 0};
    TrainBufs b;
    return carve_train(sc, w, n_img, b);
}

EC_API int ec_vit_train_forward(const ec_vit_weights *w, const void *patches, int n_img, float *feats, void *workspace,
                                size_t workspace_bytes, ec_stream_t stream)
{
    EC_TRY(check_geometry(w, "ec_vit_train_forward"));
    EC_REQUIRE(n_img > 0, "ec_vit_train_forward: n_img=%d", n_img);
    EC_REQUIRE(patches && feats && workspace, "ec_vit_train_forward: null buffer");
    const int g = w->image_size / w->patch, G = g * g, S = G + 1, W = w->width, dt = w->dtype, L = w->layers;
    const int M = n_img * S;
    Scratch sc{(unsigned char *)workspace, 0, workspace_bytes};
    TrainBufs b;
    const size_t need = carve_train(sc, w, n_img, b);
    if (need > workspace_bytes)
        return ec::fail(EC_ERR_WORKSPACE, "ec_vit_train_forward: workspace %zu < %zu bytes", workspace_bytes, need);
    float *patch_out = (float *)b.g16;
    EC_TRY(patch_embed(w, patches, n_img * G, patch_out, stream));
    EC_TRY(ec_vit_embed_train(patch_out, w->cls, w->pos, w->ln_pre_g, w->ln_pre_b, n_img, S, W, LN_EPS, b.x[0], b.pre,
                              stream));
    for (int l = 0; l < L; l++) {
        const ec_block_weights &p = w->blocks[l];
        EC_TRY(ec_layernorm(b.x[l], W, nullptr, p.ln1_g, p.ln1_b, M, W, LN_EPS, b.h1[l], W, dt, stream));
        EC_TRY(gemm_x(M, 3 * W, W, dt, EC_EPI_STORE16, b.h1[l], p.qkv_w, b.qkv[l], nullptr, nullptr, p.qkv_b, stream));
        EC_TRY(ec_attention_train(b.qkv[l], b.att[l], b.lse[l], n_img, S, W, w->heads, dt, stream));
        EC_TRY(gemm_x(M, W, W, dt, EC_EPI_RESID32, b.att[l], p.out_w, b.xm[l], b.x[l], nullptr, p.out_b, stream));
        EC_TRY(ec_layernorm(b.xm[l], W, nullptr, p.ln2_g, p.ln2_b, M, W, LN_EPS, b.h2[l], W, dt, stream));
        EC_TRY(gemm_x(M, 4 * W, W, dt, EC_EPI_GELU16_SAVE, b.h2[l], p.fc1_w, b.gact[l], nullptr, b.u[l], p.fc1_b, stream));
        EC_TRY(gemm_rows32(M, W, 4 * W, dt, EC_EPI_RESID32, b.gact[l], p.fc2_w, b.x[l + 1], b.xm[l], p.fc2_b, b.part,
                           b.part_floats, stream));
    }
    EC_TRY(ec_layernorm_split(b.x[L], (long)S * W, nullptr, w->ln_post_g, w->ln_post_b, n_img, W, LN_EPS, b.cls_hi,
                              b.cls_lo, W, dt, stream));
    return gemm3(n_img, w->out_dim, W, dt, false, b.cls_hi, b.cls_lo, w->proj_w, w->proj_w_lo, nullptr, feats, stream);
}

EC_API int ec_vit_train_backward(const ec_vit_weights *w, const ec_vit_train_weights *wt, const void *patches, int n_img,
                                 const float *d_feats, const ec_vit_grads *gr, const ec_vit_lora *lora, void *workspace,
                                 size_t workspace_bytes, ec_stream_t stream)
{
    return ec_vit_train_backward_stages(w, wt, patches, n_img, d_feats, gr, lora, 0, w ? w->layers + 2 : 0, workspace,
                                        workspace_bytes, stream);
}

// Stages: 0 = the head (ln_post, proj), 1 .. L = blocks L - 1 .. 0, L + 1 = the embedding.  The state between
// stages (the residual-stream gradient and its 16-bit copy) lives in the workspace, so a caller can run a few
// blocks, hand their finished gradients to a collective on another stream, and carry on.
EC_API int ec_vit_train_backward_stages(const ec_vit_weights *w, const ec_vit_train_weights *wt, const void *patches,
                                        int n_img, const float *d_feats, const ec_vit_grads *gr, const ec_vit_lora *lora,
                                        int stage_begin, int stage_end, void *workspace, size_t workspace_bytes,
                                        ec_stream_t stream)
{
    EC_TRY(check_geometry(w, "ec_vit_train_backward"));
    EC_REQUIRE(stage_begin >= 0 && stage_begin <= stage_end && stage_end <= w->layers + 2,
               "ec_vit_train_backward: stages %d .. %d of %d", stage_begin, stage_end, w->layers + 2);
    EC_REQUIRE(n_img > 0, "ec_vit_train_backward: n_img=%d", n_img);
    EC_REQUIRE(wt && wt->blocks && wt->proj && gr && gr->blocks, "ec_vit_train_backward: null weight / gradient structs");
    EC_REQUIRE(patches && d_feats && workspace, "ec_vit_train_backward: null buffer");
    EC_REQUIRE(!lora || (lora->blocks && lora->rank > 0 && lora->rank <= 64), "ec_vit_train_backward: LoRA rank %d (1 .. 64)",
               lora ? lora->rank : 0);
    const int g = w->image_size / w->patch, G = g * g, S = G + 1, W = w->width, dt = w->dtype, L = w->layers;
    const int M = n_img * S, D = w->out_dim;
    auto has_lora = [&](int l, int p) { return lora && lora->blocks[l].d_up[p] != nullptr; };
    Scratch sc{(unsigned char *)workspace, 0, workspace_bytes};
    TrainBufs b;
    const size_t need = carve_train(sc, w, n_img, b);
    if (need > workspace_bytes)
        return ec::fail(EC_ERR_WORKSPACE, "ec_vit_train_backward: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int Mp = b.Mp;

    // how far down the gradient has to travel: the lowest block with a gradient asked for, or the
    // embedding (-1); nothing below that is computed
    int lowest = L;
    if (gr->conv_w || gr->cls || gr->pos || gr->ln_pre_g || gr->ln_pre_b) lowest = -1;
    else
        for (int l = 0; l < L; l++) {
            const ec_block_grads &q = gr->blocks[l];
            if (q.ln1_g || q.ln1_b || q.qkv_w || q.qkv_b || q.out_w || q.out_b || q.ln2_g || q.ln2_b || q.fc1_w ||
                q.fc1_b || q.fc2_w || q.fc2_b || has_lora(l, 0) || has_lora(l, 1) || has_lora(l, 2) || has_lora(l, 3)) {
                lowest = l;
                break;
            }
        }

    // ---- head: feats = ln_post(x[:, 0]) @ proj ----
    const long ldc = (long)S * W;
    const bool below_head = lowest < L || gr->ln_post_g || gr->ln_post_b;
    if (stage_begin == 0 && stage_end > 0) {
        hipLaunchKernelGGL(ln_f32_kernel, dim3((unsigned)((n_img + 3) / 4)), dim3(256), 0, s, b.x[L], ldc, w->ln_post_g,
                           w->ln_post_b, n_img, W, LN_EPS, b.clsln, (long)W);
        EC_CHECK_HIP(hipGetLastError());
        if (gr->proj)   // d proj[W, D] = cls_ln^T . d_feats
            EC_TRY(ec_sgemm(b.clsln, 1, W, d_feats, D, 1, W, D, n_img, 1.f, 0.f, gr->proj, D, stream));
        if (below_head) {
            // d cls_ln[n, W] = d_feats . proj^T
            EC_TRY(ec_sgemm(d_feats, D, 1, wt->proj, 1, D, n_img, W, D, 1.f, 0.f, b.dclsln, W, stream));
            // the residual-stream gradient starts as zero except for the class rows; its 16-bit copy (the dX GEMMs'
            // operand) is written by whichever kernel last touched dx
            EC_CHECK_HIP(hipMemsetAsync(b.dx, 0, (size_t)M * W * 4, s));
            EC_CHECK_HIP(hipMemsetAsync(b.dx16, 0, (size_t)M * W * 2, s));
            EC_TRY(ln_backward(b.x[L], ldc, b.dclsln, W, w->ln_post_g, n_img, W, b.dx, ldc, 0, gr->ln_post_g,
                               gr->ln_post_b, b.lnpart, b.ln_wgs, s, b.dx16, dt));
        }
    }
    if (!below_head) return EC_OK;

    // ---- blocks, last to first; b.dx = d loss / d x[l + 1] on entry ----
    for (int l = L - 1; l >= 0 && l >= lowest; l--) {
        if (L - l < stage_begin) continue;
        if (L - l >= stage_end) return EC_OK;
        const ec_block_weights &p = w->blocks[l];
        const ec_block_weights_t &pt = wt->blocks[l];
        const ec_block_grads &q = gr->blocks[l];
        EC_REQUIRE(pt.qkv_wt && pt.out_wt && pt.fc1_wt && pt.fc2_wt, "ec_vit_train_backward: block %d lacks transposed weights", l);
        // x[l + 1] = xm + c_proj(QuickGELU(c_fc(ln_2(xm))))
        // (b.dx16 is the 16-bit copy of b.dx: written by the LayerNorm backward that produced it)
        if (q.fc2_b) EC_TRY(bias_grad<float>(b.dx, W, M, W, b, q.fc2_b, s));
        if (q.fc2_w) EC_TRY(weight_grad_rows(dt, b.dx16, W, b.gact[l], 4L * W, W, 4 * W, M, b, q.fc2_w, stream));
        EC_TRY(gemm_x(M, 4 * W, W, dt, EC_EPI_GELU_BWD16, b.dx16, pt.fc2_wt, b.g16, nullptr, b.u[l], nullptr, stream));
        if (q.fc1_b) {
            if (dt == EC_F16) EC_TRY(bias_grad<_Float16>((const _Float16 *)b.g16, 4L * W, M, 4 * W, b, q.fc1_b, s));
            else EC_TRY(bias_grad<__bf16>((const __bf16 *)b.g16, 4L * W, M, 4 * W, b, q.fc1_b, s));
        }
        if (q.fc1_w) EC_TRY(weight_grad_rows(dt, b.g16, 4L * W, b.h2[l], W, 4 * W, W, M, b, q.fc1_w, stream));
        EC_TRY(gemm_rows32(M, W, 4 * W, dt, EC_EPI_STORE32, b.g16, pt.fc1_wt, b.dh32, nullptr, nullptr, b.part,
                           b.part_floats, stream));
        EC_TRY(ln_backward(b.xm[l], W, b.dh32, W, p.ln2_g, M, W, b.dx, W, 1, q.ln2_g, q.ln2_b, b.lnpart, b.ln_wgs, s, b.dx16,
                           dt));
        // xm = x[l] + out_proj(attention(in_proj(ln_1(x[l]))))
        if (q.out_b) EC_TRY(bias_grad<float>(b.dx, W, M, W, b, q.out_b, s));
        if (q.out_w) EC_TRY(weight_grad_rows(dt, b.dx16, W, b.att[l], W, W, W, M, b, q.out_w, stream));
        EC_TRY(gemm_x(M, W, W, dt, EC_EPI_STORE16, b.dx16, pt.out_wt, b.da16, nullptr, nullptr, nullptr, stream));
        EC_TRY(ec_attention_backward(b.qkv[l], b.att[l], b.lse[l], b.da16, b.g16, b.delta, n_img, S, W, w->heads, dt,
                                     stream));
        if (q.qkv_b) {
            if (dt == EC_F16) EC_TRY(bias_grad<_Float16>((const _Float16 *)b.g16, 3L * W, M, 3 * W, b, q.qkv_b, s));
            else EC_TRY(bias_grad<__bf16>((const __bf16 *)b.g16, 3L * W, M, 3 * W, b, q.qkv_b, s));
        }
        const bool lora_qkv = has_lora(l, 0) || has_lora(l, 1) || has_lora(l, 2);
        if (q.qkv_w) EC_TRY(weight_grad_rows(dt, b.g16, 3L * W, b.h1[l], W, 3 * W, W, M, b, q.qkv_w, stream));
        if (lora_qkv || has_lora(l, 3)) {
            // q, k, v multiply ln_1(x) and receive dq | dk | dv; out_proj multiplies the attention output and
            // receives the residual gradient of xm (its 16-bit copy is still in dx16)
            const ec_block_lora &lb = lora->blocks[l];
            LoraJob jobs[4];
            int n = 0;
            const size_t esz = 2;
            for (int pj = 0; pj < 4; pj++) {
                if (!has_lora(l, pj)) continue;
                EC_REQUIRE(lb.down16[pj] && lb.up16_t[pj] && lb.d_down[pj], "ec_vit_train_backward: block %d LoRA item %d incomplete", l, pj);
                LoraJob &j = jobs[n++];
                if (pj < 3) {
                    j.x = b.h1[l], j.ldx = W;
                    j.dy = static_cast<const unsigned char *>(b.g16) + (size_t)pj * W * esz, j.ldy = 3L * W;
                } else {
                    j.x = b.att[l], j.ldx = W, j.dy = b.dx16, j.ldy = W;
                }
                j.down16 = lb.down16[pj], j.up16t = lb.up16_t[pj], j.d_up = lb.d_up[pj], j.d_down = lb.d_down[pj];
            }
            EC_TRY(lora_grads(dt, jobs, n, M, W, lora->rank, b, s));
        }
        if (l == lowest && !q.ln1_g && !q.ln1_b) break;   // nothing below needs d x[l]
        EC_TRY(gemm_rows32(M, W, 3 * W, dt, EC_EPI_STORE32, b.g16, pt.qkv_wt, b.dh32, nullptr, nullptr, b.part,
                           b.part_floats, stream));
        EC_TRY(ln_backward(b.x[l], W, b.dh32, W, p.ln1_g, M, W, b.dx, W, 1, q.ln1_g, q.ln1_b, b.lnpart, b.ln_wgs, s, b.dx16,
                           dt));
    }
    if (lowest >= 0 || stage_end <= L + 1) return EC_OK;

    // ---- embedding: x[0] = ln_pre([cls; patches . conv1^T] + pos) ----
    EC_TRY(ln_backward(b.pre, W, b.dx, W, w->ln_pre_g, M, W, b.dh32, W, 0, gr->ln_pre_g, gr->ln_pre_b, b.lnpart, b.ln_wgs,
                       s));
    if (gr->pos || gr->cls) {
        hipLaunchKernelGGL(pos_grad_kernel, dim3((unsigned)S), dim3(256), 0, s, b.dh32, n_img, S, W, gr->pos, gr->cls);
        EC_CHECK_HIP(hipGetLastError());
    }
    if (gr->conv_w) {
        const int R = n_img * G;
        EC_TRY(transpose<2>(dt, b.dh32, W, R, W, Mp, G, S, 1, b.ta, nullptr, s));
        EC_TRY(transpose<0>(dt, patches, w->kpad, R, w->kpad, Mp, 0, 0, 0, b.tb, nullptr, s));
        EC_TRY(weight_grad(dt, b.ta, b.tb, W, w->kpad, b, gr->conv_w, 3 * w->patch * w->patch, stream));
    }
    return EC_OK;
}

EC_API int ec_pack_weight16_batched(const ec_pack_item *items, int n_items, int rows, int cols, int dtype,
                                    ec_stream_t stream)
{
    EC_REQUIRE(n_items > 0 && rows > 0 && cols > 0, "ec_pack_weight16_batched: %d items of %d x %d", n_items, rows, cols);
    EC_REQUIRE(items, "ec_pack_weight16_batched: null item table");
    const dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64), (unsigned)n_items);
    hipStream_t s = static_cast<hipStream_t>(stream);
    ec::ProfScope prof(ec::PROF_PACK, s, 0, (double)rows * cols * 10.0 * n_items);
    if (dtype == EC_F16)
        hipLaunchKernelGGL(pack_weight_kernel<EC_F16>, grid, dim3(256), 0, s, items, rows, cols);
    else if (dtype == EC_BF16)
        hipLaunchKernelGGL(pack_weight_kernel<EC_BF16>, grid, dim3(256), 0, s, items, rows, cols);
    else
        return ec::fail(EC_ERR_INVALID, "ec_pack_weight16_batched: unknown dtype %d", dtype);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

EC_API size_t ec_layernorm_backward_partials(int rows, int width)
{
    if (rows <= 0 || width <= 0) return 0;
    const int wgs = (rows + 3) / 4 < 1024 ? (rows + 3) / 4 : 1024;
    return (size_t)wgs * 2 * width;
}

EC_API int ec_layernorm_backward(const float *x, long ldx, const float *dy, long ldy, const float *gamma, int rows,
                                 int width, float eps, float *dx, long ldo, int accumulate, float *d_gamma, float *d_beta,
                                 float *partials, ec_stream_t stream)
{
    EC_REQUIRE(rows >= 0 && width > 0 && width % 4 == 0 && width <= LN_MAXV * 256,
               "ec_layernorm_backward: width=%d must be a multiple of 4 and <= %d", width, LN_MAXV * 256);
    if (rows == 0) return EC_OK;
    EC_REQUIRE(x && dy && gamma && dx, "ec_layernorm_backward: null buffer");
    EC_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && ldo % 4 == 0, "ec_layernorm_backward: strides must be multiples of 4");
    EC_REQUIRE(!(d_gamma || d_beta) || partials, "ec_layernorm_backward: d_gamma / d_beta need the partials scratch");
    EC_REQUIRE(eps == LN_EPS, "ec_layernorm_backward: eps is fixed at 1e-5");
    return ln_backward(x, ldx, dy, ldy, gamma, rows, width, dx, ldo, accumulate, d_gamma, d_beta, partials, 1024,
                       static_cast<hipStream_t>(stream));
}

EC_API int ec_lora_merge_batched(const ec_lora_item *items, int n_items, int rows, int cols, int r, ec_stream_t stream)
{
    EC_REQUIRE(n_items > 0 && rows > 0 && cols > 0 && cols % 4 == 0 && r > 0 && r <= LORA_MAXR,
               "ec_lora_merge_batched: %d items of %d x %d, r = %d (<= %d)", n_items, rows, cols, r, LORA_MAXR);
    EC_REQUIRE(items, "ec_lora_merge_batched: null item table");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ec::ProfScope prof(ec::PROF_SGEMM, s, 2.0 * rows * cols * r * n_items, 8.0 * rows * cols * n_items);
    hipLaunchKernelGGL(lora_merge_kernel, dim3((unsigned)((cols / 4 + 255) / 256), (unsigned)((rows + 15) / 16), (unsigned)n_items),
                       dim3(256), 0, s, items, rows, cols, r);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

EC_API size_t ec_lora_grad_scratch_floats(int n_items, int rows, int cols, int r)
{
    if (n_items <= 0 || rows <= 0 || cols <= 0 || r <= 0) return 0;
    return (size_t)n_items * 16 * r * cols;
}

EC_API int ec_lora_grad_batched(const ec_lora_item *items, int n_items, int rows, int cols, int r, float *scratch,
                                ec_stream_t stream)
{
    EC_REQUIRE(n_items > 0 && rows > 0 && cols > 0 && cols % 4 == 0 && r > 0 && r <= LORA_MAXR,
               "ec_lora_grad_batched: %d items of %d x %d, r = %d (<= %d)", n_items, rows, cols, r, LORA_MAXR);
    EC_REQUIRE(items && scratch, "ec_lora_grad_batched: null buffer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int slab = (rows + 15) / 16, slabs = (rows + slab - 1) / slab;   // 16 slabs: the scratch's size
    const long n = (long)r * cols;
    ec::ProfScope prof(ec::PROF_SGEMM, s, 4.0 * rows * cols * r * n_items, 8.0 * rows * cols * n_items);
    hipLaunchKernelGGL(lora_dup_kernel, dim3((unsigned)((rows + 15) / 16), 1, (unsigned)n_items), dim3(256), 0, s, items, rows,
                       cols, r);
    hipLaunchKernelGGL(lora_ddown_kernel, dim3((unsigned)((cols / 4 + 255) / 256), (unsigned)slabs, (unsigned)n_items), dim3(256), 0,
                       s, items, rows, cols, r, slab, scratch);
    hipLaunchKernelGGL(lora_ddown_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256), (unsigned)n_items), dim3(256), 0, s, items,
                       scratch, slabs, n);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

EC_API int ec_adam_step_multi(const ec_adam_item *items, int n_items, int64_t max_n, float lr0, float lr1, float beta1,
                              float beta2, float eps, float weight_decay, int step, const int32_t *skip_flag,
                              const float *step_scalars, ec_stream_t stream)
{
    EC_REQUIRE(n_items >= 0 && max_n >= 0 && (step >= 1 || step_scalars), "ec_adam_step_multi: n_items=%d step=%d", n_items, step);
    if (step < 1) step = 1;      // the bias corrections come from step_scalars
    if (n_items == 0 || max_n == 0) return EC_OK;
    EC_REQUIRE(items, "ec_adam_step_multi: null item table");
    const double bc1 = 1.0 - __builtin_pow((double)beta1, (double)step);
    const double bc2 = 1.0 - __builtin_pow((double)beta2, (double)step);
    const long blocks = ((long)max_n + 255) / 256;
    hipStream_t s = static_cast<hipStream_t>(stream);
    ec::ProfScope prof(ec::PROF_OPTIMIZER, s, 0, 0);
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)(blocks < 512 ? blocks : 512), (unsigned)n_items), dim3(256), 0, s,
                       items, lr0, lr1, beta1, beta2, eps, weight_decay, (float)bc1, (float)__builtin_sqrt(bc2), skip_flag, step_scalars);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

EC_API int ec_grad_unscale_check(float *grad, int64_t n, float inv_scale, int32_t *found_inf, const float *step_scalars,
                                 ec_stream_t stream)
{
    EC_REQUIRE(n >= 0, "ec_grad_unscale_check: n=%lld", (long long)n);
    if (n == 0) return EC_OK;
    EC_REQUIRE(grad && found_inf, "ec_grad_unscale_check: null buffer");
    const long blocks = (n + 255) / 256;
    ec::ProfScope prof(ec::PROF_OPTIMIZER, static_cast<hipStream_t>(stream), 0, 8.0 * n);
    hipLaunchKernelGGL(unscale_check_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), grad, (long)n, inv_scale, found_inf, step_scalars);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

EC_API int ec_ft_loss_grad(const float *img_feats, const int32_t *row_idx, const uint8_t *valid, const int32_t *labels,
                           const float *text_param, int B, int T, int D, int K, float logit_scale, int agg,
                           int use_probs_loss, float grad_scale, const float *step_scalars,
                           float *loss, float *grad_text, float *grad_img, float *agg_logits, void *workspace,
                           size_t workspace_bytes, ec_stream_t stream)
{
    EC_REQUIRE(B > 0 && T > 0 && D > 0 && K > 0, "ec_ft_loss_grad: bad shape");
    EC_REQUIRE(img_feats && valid && labels && text_param && loss && grad_img && workspace, "ec_ft_loss_grad: null buffer");
    const size_t base = ec_fs_text_train_workspace_bytes(B, T, D, K);
    const size_t R = (size_t)B * T;
    EC_REQUIRE(workspace_bytes >= base + (R > (size_t)K ? R : (size_t)K) * D * 4, "ec_ft_loss_grad: workspace too small");
    // the few-shot head leaves Fn (normalised views), dL (per-view logit gradients) and u (normalised text) at
    // the front of its workspace, in this order (train.hip: ec_fs_text_loss_grad)
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    float *gt = grad_text ? grad_text : reinterpret_cast<float *>(ws + base);   // unused text gradient: scratch
    EC_TRY(ec::fs_text_loss_grad(img_feats, row_idx, valid, labels, text_param, B, T, D, K, logit_scale, agg, use_probs_loss,
                                 loss, gt, agg_logits, workspace, base, stream));
    auto a256 = [](size_t x) { return (x + 255) / 256 * 256; };
    const float *Fn = reinterpret_cast<const float *>(ws);
    const float *dL = reinterpret_cast<const float *>(ws + a256(R * D * 4));
    const float *u = reinterpret_cast<const float *>(ws + a256(R * D * 4) + a256(R * K * 4));
    float *dfn = reinterpret_cast<float *>(ws + base);
    // dFn[R, D] = logit_scale * dL[R, K] . u[K, D]
    EC_TRY(ec_sgemm(dL, K, 1, u, D, 1, (int)R, D, K, logit_scale, 0.f, dfn, D, stream));
    hipLaunchKernelGGL(feat_grad_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       img_feats, Fn, dfn, valid, row_idx, (int)R, D, grad_scale, grad_img, step_scalars);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

}  // extern "C"
