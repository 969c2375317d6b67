// LayerNorm kernels of the CLIP towers (gfx950): HBM-bound, one wave per row.
//
//  * ec_layernorm: fp32 row -> LayerNorm (fp32 statistics, eps inside the sqrt,
//    as the fp32-computing LayerNorm subclass of un-vendored openai/CLIP
//    clip/model.py; call sites models/clip_cls.py:84,101) -> 16-bit GEMM operand.
//    Row stride is free, so ln_post reads the CLS rows in place.
//  * ec_vit_embed: x[n, 0] = class_embedding, x[n, 1+p] = patch GEMM row, plus
//    positional_embedding, then ln_pre -> the fp32 residual stream.
//  * ec_text_embed: token_embedding[token] + positional_embedding -> residual stream.
//
// Each lane keeps its slice of the row in registers (float4 x up to 8), two
// shuffle reductions (mean, then centred variance), one write.
#include "common.h"
#include "mfma.h"
#include "tower_ops.h"

namespace {

using namespace ec;

constexpr int LN_MAXV = 8;  // float4 per lane: width <= 2048

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// normalises the lane-resident row in place: v = (v - mean) * rstd * gamma + beta
__device__ __forceinline__ void ln_row(float4 (&v)[LN_MAXV], int nv, int width, int lane,
                                       const float *gamma, const float *beta, float eps)
{
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; i++)
        if (i < nv) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
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
            v[i].x = (v[i].x - mean) * rstd * g.x + b.x;
            v[i].y = (v[i].y - mean) * rstd * g.y + b.y;
            v[i].z = (v[i].z - mean) * rstd * g.z + b.z;
            v[i].w = (v[i].w - mean) * rstd * g.w + b.w;
        }
}

// lanes cover the row in float4 units: unit u = i*64 + lane, valid if u*4 < width
__device__ __forceinline__ int units(int width, int lane)
{
    const int total = width / 4;
    return (total - lane + 63) / 64;  // number of i with i*64 + lane < total
}

// HLIN: the row arrives as the two planes of the folded chain's residual stream (x = hi plane in DT, x_lo = fp16 lo plane,
// both at row stride ldx) and is joined to fp32 on the way in: the LayerNorm of the split-operand blocks
// four fp32 -> four OCP e4m3 bytes (round to nearest even; clamped to +-448 first: e4m3fn has no infinity)
__device__ __forceinline__ unsigned pack_e4m3(float a, float b, float c, float d, float scale)
{
    a = __builtin_amdgcn_fmed3f(a * scale, -448.f, 448.f), b = __builtin_amdgcn_fmed3f(b * scale, -448.f, 448.f);
    c = __builtin_amdgcn_fmed3f(c * scale, -448.f, 448.f), d = __builtin_amdgcn_fmed3f(d * scale, -448.f, 448.f);
    unsigned r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0u, false);
    return __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
}

// F8 (ec_layernorm_hl8): the lo part leaves as e4m3 of lo x lo_scale -- the A_lo8 operand of ec_gemm -- and, with out_hi8,
// an e4m3 copy of the hi part x hi_scale (A8): one byte per element at the SAME byte row pitch as the 16-bit output (the
// first `width` bytes of each 2 ldo-byte row; ec_gemm_args.A_lo8)
template <int DT, bool HLIN = false, bool F8 = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const void *x_v, const _Float16 *x_lo, long ldx,
                                                        const int *row_idx, const float *gamma,
                                                        const float *beta, int rows, int width,
                                                        float eps, void *out, long ldo, void *out_lo,
                                                        void *out_hi8 = nullptr, float lo_scale = 1.f, float hi_scale = 1.f)
{
    typedef typename T16<DT>::elem elem;
    typedef typename T16<DT>::v4 v4;
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = units(width, lane);
    float4 v[LN_MAXV];
    const long src = (row_idx ? (long)row_idx[row] : row) * ldx;
#pragma unroll
    for (int i = 0; i < LN_MAXV; i++)
        if (i < nv) {
            if constexpr (HLIN) {
                const v4 h = *reinterpret_cast<const v4 *>(static_cast<const elem *>(x_v) + src + (i * 64 + lane) * 4);
                const f16x4 l = *reinterpret_cast<const f16x4 *>(x_lo + src + (i * 64 + lane) * 4);
                v[i] = make_float4((float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2],
                                   (float)h[3] + (float)l[3]);
            } else {
                v[i] = *reinterpret_cast<const float4 *>(static_cast<const float *>(x_v) + src + (i * 64 + lane) * 4);
            }
        }
    ln_row(v, nv, width, lane, gamma, beta, eps);
    elem *o = (elem *)out + row * ldo;
#pragma unroll
    for (int i = 0; i < LN_MAXV; i++)
        if (i < nv) {
            v4 p = {to16(v[i].x, elem()), to16(v[i].y, elem()), to16(v[i].z, elem()),
                    to16(v[i].w, elem())};
            *reinterpret_cast<v4 *>(o + (i * 64 + lane) * 4) = p;
            if constexpr (F8) {
                unsigned char *lo8 = static_cast<unsigned char *>(out_lo) + row * ldo * 2 + (i * 64 + lane) * 4;
                *reinterpret_cast<unsigned *>(lo8) = pack_e4m3(v[i].x - (float)p[0], v[i].y - (float)p[1], v[i].z - (float)p[2],
                                                               v[i].w - (float)p[3], lo_scale);
                if (out_hi8) {
                    unsigned char *hi8 = static_cast<unsigned char *>(out_hi8) + row * ldo * 2 + (i * 64 + lane) * 4;
                    *reinterpret_cast<unsigned *>(hi8) = pack_e4m3((float)p[0], (float)p[1], (float)p[2], (float)p[3], hi_scale);
                }
            } else if (out_lo) {   // split precision: lo = 16-bit(x - hi), x ~ hi + lo to ~2^-22
                v4 q = {to16(v[i].x - (float)p[0], elem()), to16(v[i].y - (float)p[1], elem()),
                        to16(v[i].z - (float)p[2], elem()), to16(v[i].w - (float)p[3], elem())};
                *reinterpret_cast<v4 *>((elem *)out_lo + row * ldo + (i * 64 + lane) * 4) = q;
            }
        }
}

// LayerNorm statistics of 16-bit rows (the hi plane of the residual stream): one wave per row, the row in
// registers as eight-element chunks, mean and centred variance like ln_row -> (rstd, -rstd * mean)
template <int DT>
__global__ __launch_bounds__(256) void row_stats_kernel(const void *x, long ldx, int rows, int width, float eps,
                                                        float *stats)
{
    typedef typename T16<DT>::elem elem;
    typedef typename T16<DT>::v8 v8;
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const elem *xr = (const elem *)x + row * ldx;
    constexpr int MAXC = LN_MAXV / 2;          // 8-element chunks per lane: width <= 2048
    const int total = width / 8, nc = (total - lane + 63) / 64;
    v8 v[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; i++)
        if (i < nc) v[i] = *reinterpret_cast<const v8 *>(xr + (i * 64 + lane) * 8);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; i++)
        if (i < nc) {
#pragma unroll
            for (int e = 0; e < 8; e++) s += (float)v[i][e];
        }
    const float mean = wave_sum(s) / (float)width;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; i++)
        if (i < nc) {
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float d = (float)v[i][e] - mean;
                q += d * d;
            }
        }
    const float rstd = 1.f / __builtin_sqrtf(wave_sum(q) / (float)width + eps);
    if (lane == 0) *reinterpret_cast<float2 *>(stats + 2 * row) = make_float2(rstd, -rstd * mean);
}

// per-group (sum, sum of squares) of a row's hi values (EC_EPI_RESID_HL, ec_gemm_args.row_sums) -> (rstd, -rstd mean)
__global__ __launch_bounds__(256) void row_stats_merge_kernel(const float *sums, int rows, int groups, int width, float eps,
                                                              float *stats)
{
    // sixteen lanes per row: a row's groups are contiguous (128 B at 16 groups), one coalesced 8-byte load per lane
    const int sub = threadIdx.x & 15;
    const long row = ((long)blockIdx.x * 256 + threadIdx.x) >> 4;
    float s = 0.f, q = 0.f;
    if (row < rows) {
        const float2 *p = reinterpret_cast<const float2 *>(sums) + row * groups;
        for (int i = sub; i < groups; i += 16) {
            const float2 v = p[i];
            s += v.x, q += v.y;
        }
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        s += __shfl_xor(s, o, 16);
        q += __shfl_xor(q, o, 16);
    }
    if (row < rows && sub == 0) {
        const float mean = s / (float)width;
        const float var = fmaxf(q / (float)width - mean * mean, 0.f);
        const float rstd = 1.f / __builtin_sqrtf(var + eps);
        *reinterpret_cast<float2 *>(stats + 2 * row) = make_float2(rstd, -rstd * mean);
    }
}

// fp32 [n] -> (optionally QuickGELU) -> 16-bit hi and lo parts.  LO_F16: the lo part as fp16 whatever DT (the planes of
// the folded chain's residual stream: EC_EPI_RESID_HL reads its lo plane as fp16)
template <int DT, bool LO_F16 = false>
__global__ __launch_bounds__(256) void split16_kernel(const float *x, long n, int gelu, void *hi,
                                                      void *lo)
{
    typedef typename T16<DT>::elem elem;
    typedef typename T16<DT>::v4 v4;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        float4 v = *reinterpret_cast<const float4 *>(x + i);
        if (gelu) {
            v.x = v.x / (1.f + expf(-1.702f * v.x));
            v.y = v.y / (1.f + expf(-1.702f * v.y));
            v.z = v.z / (1.f + expf(-1.702f * v.z));
            v.w = v.w / (1.f + expf(-1.702f * v.w));
        }
        v4 p = {to16(v.x, elem()), to16(v.y, elem()), to16(v.z, elem()), to16(v.w, elem())};
        *reinterpret_cast<v4 *>((elem *)hi + i) = p;
        if constexpr (LO_F16) {
            const f16x4 q = {(_Float16)(v.x - (float)p[0]), (_Float16)(v.y - (float)p[1]), (_Float16)(v.z - (float)p[2]),
                             (_Float16)(v.w - (float)p[3])};
            *reinterpret_cast<f16x4 *>((_Float16 *)lo + i) = q;
        } else {
            const v4 q = {to16(v.x - (float)p[0], elem()), to16(v.y - (float)p[1], elem()),
                          to16(v.z - (float)p[2], elem()), to16(v.w - (float)p[3], elem())};
            *reinterpret_cast<v4 *>((elem *)lo + i) = q;
        }
    }
}

// fp32 in -> fp32 out with an additive table (positional embedding) and a row
// source that is either a broadcast vector (class token) or a GEMM output row.
template <int DT = -1>   // DT >= 0: the residual stream goes out as hi (DT) + lo (fp16) planes (x = hi plane pointer)
__global__ __launch_bounds__(256) void vit_embed_kernel(const float *patch, const float *cls,
                                                        const float *pos, const float *gamma,
                                                        const float *beta, int n_img, int seq,
                                                        int width, float eps, float *x, float *pre, _Float16 *x_lo = nullptr)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)n_img * seq) return;
    const int n = (int)(row / seq), s = (int)(row % seq);
    const float *src = s == 0 ? cls : patch + ((long)n * (seq - 1) + (s - 1)) * width;
    const float *pr = pos + (long)s * width;
    const int nv = units(width, lane);
    float4 v[LN_MAXV];
#pragma unroll
    for (int i = 0; i < LN_MAXV; i++)
        if (i < nv) {
            const int c = (i * 64 + lane) * 4;
            const float4 a = *reinterpret_cast<const float4 *>(src + c);
            const float4 p = *reinterpret_cast<const float4 *>(pr + c);
            v[i] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
            // training keeps the un-normalised embedding: ln_pre's backward pass needs it
            if (pre) *reinterpret_cast<float4 *>(pre + row * width + c) = v[i];
        }
    ln_row(v, nv, width, lane, gamma, beta, eps);
    if constexpr (DT >= 0) {
        typedef typename T16<DT < 0 ? 0 : DT>::elem elem;
        typedef typename T16<DT < 0 ? 0 : DT>::v4 v4;
        elem *oh = reinterpret_cast<elem *>(x) + row * width;
        _Float16 *ol = x_lo + row * width;
#pragma unroll
        for (int i = 0; i < LN_MAXV; i++)
            if (i < nv) {
                const v4 h = {to16(v[i].x, elem()), to16(v[i].y, elem()), to16(v[i].z, elem()), to16(v[i].w, elem())};
                const f16x4 l = {(_Float16)(v[i].x - (float)h[0]), (_Float16)(v[i].y - (float)h[1]),
                                 (_Float16)(v[i].z - (float)h[2]), (_Float16)(v[i].w - (float)h[3])};
                *reinterpret_cast<v4 *>(oh + (i * 64 + lane) * 4) = h;
                *reinterpret_cast<f16x4 *>(ol + (i * 64 + lane) * 4) = l;
            }
        return;
    }
    float *o = x + row * width;
#pragma unroll
    for (int i = 0; i < LN_MAXV; i++)
        if (i < nv) *reinterpret_cast<float4 *>(o + (i * 64 + lane) * 4) = v[i];
}

// rows of a residual stream kept as hi + lo planes (row stride ld elements) -> fp32 [rows, width]
template <int DT>
__global__ __launch_bounds__(256) void join_hl_kernel(const void *x_hi, const _Float16 *x_lo, long ld, int rows, int width,
                                                      float *out)
{
    typedef typename T16<DT>::elem elem;
    const long n4 = (long)rows * (width / 4);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long r = i / (width / 4), c = (i % (width / 4)) * 4;
        const elem *h = (const elem *)x_hi + r * ld + c;
        const _Float16 *l = x_lo + r * ld + c;
        *reinterpret_cast<float4 *>(out + r * width + c) = make_float4((float)h[0] + (float)l[0], (float)h[1] + (float)l[1],
                                                                       (float)h[2] + (float)l[2], (float)h[3] + (float)l[3]);
    }
}

__global__ __launch_bounds__(256) void text_embed_kernel(const int *tokens, const float *table,
                                                         const float *pos, int n_txt, int ctx,
                                                         int width, int vocab, float *x)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)n_txt * ctx) return;
    const int s = (int)(row % ctx);
    int tok = tokens[row];
    tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
    const float *src = table + (long)tok * width, *pr = pos + (long)s * width;
    float *o = x + row * width;
    for (int c = lane * 4; c < width; c += 256) {
        const float4 a = *reinterpret_cast<const float4 *>(src + c);
        const float4 p = *reinterpret_cast<const float4 *>(pr + c);
        *reinterpret_cast<float4 *>(o + c) = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
    }
}

}  // namespace

// ---- internal entry points of the folded-LayerNorm image tower (towers.hip; declared in tower_ops.h) ----
namespace ec_tower {

int vit_embed_hl(const float *patch, const float *cls, const float *pos, const float *gamma, const float *beta,
                 int n_img, int seq, int width, float eps, void *x_hi, void *x_lo, int dtype, ec_stream_t stream)
{
    EC_REQUIRE(width % 4 == 0 && width <= LN_MAXV * 256 && seq >= 2, "vit_embed_hl: bad shape");
    if (n_img == 0) return EC_OK;
    const long rows = (long)n_img * seq;
    hipStream_t s = static_cast<hipStream_t>(stream);
    ec::ProfScope prof(ec::PROF_EMBED, s, 0, (double)rows * width * 8.0);
    const dim3 grid((unsigned)ec::ceil_div(rows, 4L));
    if (dtype == EC_F16)
        hipLaunchKernelGGL(vit_embed_kernel<EC_F16>, grid, dim3(256), 0, s, patch, cls, pos, gamma, beta, n_img, seq, width,
                           eps, (float *)x_hi, (float *)nullptr, (_Float16 *)x_lo);
    else if (dtype == EC_BF16)
        hipLaunchKernelGGL(vit_embed_kernel<EC_BF16>, grid, dim3(256), 0, s, patch, cls, pos, gamma, beta, n_img, seq, width,
                           eps, (float *)x_hi, (float *)nullptr, (_Float16 *)x_lo);
    else
        return ec::fail(EC_ERR_INVALID, "vit_embed_hl: unknown dtype %d", dtype);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

// fp32 residual stream -> the folded chain's planes: hi in `dtype`, lo ALWAYS fp16 (ec_vit_weights.precise_blocks: the
// hand-over from the split-precision blocks)
int split_hl(const float *x, long n, void *x_hi, void *x_lo, int dtype, ec_stream_t stream)
{
    EC_REQUIRE(n >= 0 && n % 4 == 0 && x && x_hi && x_lo, "split_hl: bad arguments");
    if (n == 0) return EC_OK;
    const unsigned grid = (unsigned)((n / 4 + 255) / 256 < 65536 ? (n / 4 + 255) / 256 : 65536);
    hipStream_t s = static_cast<hipStream_t>(stream);
    ec::ProfScope prof(ec::PROF_LAYERNORM, s, 0, (double)n * 8.0);
    if (dtype == EC_F16)
        hipLaunchKernelGGL((split16_kernel<EC_F16, true>), dim3(grid), dim3(256), 0, s, x, n, 0, x_hi, x_lo);
    else if (dtype == EC_BF16)
        hipLaunchKernelGGL((split16_kernel<EC_BF16, true>), dim3(grid), dim3(256), 0, s, x, n, 0, x_hi, x_lo);
    else
        return ec::fail(EC_ERR_INVALID, "split_hl: unknown dtype %d", dtype);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

int join_hl_rows(const void *x_hi, const void *x_lo, long ld, int rows, int width, float *out, int dtype, ec_stream_t stream)
{
    if (rows == 0) return EC_OK;
    EC_REQUIRE(width % 4 == 0 && ld % 4 == 0, "join_hl_rows: bad shape");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long n4 = (long)rows * (width / 4);
    const unsigned grid = (unsigned)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    if (dtype == EC_F16)
        hipLaunchKernelGGL(join_hl_kernel<EC_F16>, dim3(grid), dim3(256), 0, s, x_hi, (const _Float16 *)x_lo, ld, rows, width, out);
    else if (dtype == EC_BF16)
        hipLaunchKernelGGL(join_hl_kernel<EC_BF16>, dim3(grid), dim3(256), 0, s, x_hi, (const _Float16 *)x_lo, ld, rows, width, out);
    else
        return ec::fail(EC_ERR_INVALID, "join_hl_rows: unknown dtype %d", dtype);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

}  // namespace ec_tower

extern "C" {

EC_API int ec_layernorm(const float *x, long ldx, const int32_t *row_idx, const float *gamma,
                        const float *beta, int rows, int width, float eps, void *out16, long ldo,
                        int dtype, ec_stream_t stream)
{
    return ec_layernorm_split(x, ldx, row_idx, gamma, beta, rows, width, eps, out16, nullptr, ldo,
                              dtype, stream);
}

EC_API int ec_row_stats(const void *x16, long ldx, int rows, int width, float eps, float *stats, int dtype,
                        ec_stream_t stream)
{
    EC_REQUIRE(rows >= 0 && width > 0 && width % 8 == 0 && width <= LN_MAXV * 256,
               "ec_row_stats: width=%d must be a multiple of 8 and <= %d", width, LN_MAXV * 256);
    if (rows == 0) return EC_OK;
    EC_REQUIRE(x16 && stats && ldx % 8 == 0 && ((uintptr_t)x16 & 15) == 0, "ec_row_stats: null or misaligned buffer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ec::ProfScope prof(ec::PROF_LAYERNORM, s, 0, (double)rows * width * 2.0);
    if (dtype == EC_F16)
        hipLaunchKernelGGL(row_stats_kernel<EC_F16>, dim3(ec::ceil_div(rows, 4)), dim3(256), 0, s, x16, ldx, rows, width, eps, stats);
    else if (dtype == EC_BF16)
        hipLaunchKernelGGL(row_stats_kernel<EC_BF16>, dim3(ec::ceil_div(rows, 4)), dim3(256), 0, s, x16, ldx, rows, width, eps, stats);
    else
        return ec::fail(EC_ERR_INVALID, "ec_row_stats: unknown dtype %d", dtype);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

EC_API int ec_row_stats_merge(const float *sums, int rows, int groups, int width, float eps, float *stats,
                              ec_stream_t stream)
{
    EC_REQUIRE(rows >= 0 && groups > 0 && width == groups * 64, "ec_row_stats_merge: width %d != 64 x %d groups", width, groups);
    if (rows == 0) return EC_OK;
    EC_REQUIRE(sums && stats, "ec_row_stats_merge: null buffer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ec::ProfScope prof(ec::PROF_LAYERNORM, s, 0, (double)rows * (groups * 8.0 + 8.0));
    hipLaunchKernelGGL(row_stats_merge_kernel, dim3(ec::ceil_div(rows, 16)), dim3(256), 0, s, sums, rows, groups, width, eps,
                       stats);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

EC_API int ec_split16(const float *x, long n, int gelu, void *hi16, void *lo16, int dtype,
                      ec_stream_t stream)
{
    EC_REQUIRE(n >= 0 && n % 4 == 0, "ec_split16: n=%ld must be a multiple of 4", n);
    if (n == 0) return EC_OK;
    EC_REQUIRE(x && hi16 && lo16, "ec_split16: null buffer");
    const unsigned grid = (unsigned)((n / 4 + 255) / 256 < 65536 ? (n / 4 + 255) / 256 : 65536);
    hipStream_t s = static_cast<hipStream_t>(stream);
    ec::ProfScope prof(ec::PROF_LAYERNORM, s, 0, (double)n * 8.0);
    if (dtype == EC_F16)
        hipLaunchKernelGGL(split16_kernel<EC_F16>, dim3(grid), dim3(256), 0, s, x, n, gelu, hi16, lo16);
    else if (dtype == EC_BF16)
        hipLaunchKernelGGL(split16_kernel<EC_BF16>, dim3(grid), dim3(256), 0, s, x, n, gelu, hi16,
                           lo16);
    else
        return ec::fail(EC_ERR_INVALID, "ec_split16: unknown dtype %d", dtype);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

EC_API int ec_layernorm_split(const float *x, long ldx, const int32_t *row_idx, const float *gamma,
                              const float *beta, int rows, int width, float eps, void *out16,
                              void *out16_lo, long ldo, int dtype, ec_stream_t stream)
{
    EC_REQUIRE(rows >= 0 && width > 0 && width % 4 == 0 && width <= LN_MAXV * 256,
               "ec_layernorm: width=%d must be a multiple of 4 and <= %d", width, LN_MAXV * 256);
    if (rows == 0) return EC_OK;
    EC_REQUIRE(x && gamma && beta && out16, "ec_layernorm: null buffer");
    EC_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0, "ec_layernorm: strides must be multiples of 4");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid(ec::ceil_div(rows, 4)), block(256);
    ec::ProfScope prof(ec::PROF_LAYERNORM, s, 0, (double)rows * width * 6.0);
    if (dtype == EC_F16)
        hipLaunchKernelGGL((layernorm_kernel<EC_F16, false>), grid, block, 0, s, x, (const _Float16 *)nullptr, ldx, row_idx, gamma,
                           beta, rows, width, eps, out16, ldo, out16_lo);
    else if (dtype == EC_BF16)
        hipLaunchKernelGGL((layernorm_kernel<EC_BF16, false>), grid, block, 0, s, x, (const _Float16 *)nullptr, ldx, row_idx, gamma,
                           beta, rows, width, eps, out16, ldo, out16_lo);
    else
        return ec::fail(EC_ERR_INVALID, "ec_layernorm: unknown dtype %d", dtype);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

EC_API int ec_layernorm_hl(const void *x_hi, const void *x_lo, long ldx, const float *gamma, const float *beta, int rows,
                           int width, float eps, void *out16, void *out16_lo, long ldo, int dtype, ec_stream_t stream)
{
    EC_REQUIRE(rows >= 0 && width > 0 && width % 4 == 0 && width <= LN_MAXV * 256,
               "ec_layernorm_hl: width=%d must be a multiple of 4 and <= %d", width, LN_MAXV * 256);
    if (rows == 0) return EC_OK;
    EC_REQUIRE(x_hi && x_lo && gamma && beta && out16 && out16_lo, "ec_layernorm_hl: null buffer");
    EC_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0, "ec_layernorm_hl: strides must be multiples of 4");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid(ec::ceil_div(rows, 4)), block(256);
    ec::ProfScope prof(ec::PROF_LAYERNORM, s, 0, (double)rows * width * 8.0);
    const _Float16 *lo = static_cast<const _Float16 *>(x_lo);
    if (dtype == EC_F16)
        hipLaunchKernelGGL((layernorm_kernel<EC_F16, true>), grid, block, 0, s, x_hi, lo, ldx, (const int *)nullptr, gamma, beta,
                           rows, width, eps, out16, ldo, out16_lo);
    else if (dtype == EC_BF16)
        hipLaunchKernelGGL((layernorm_kernel<EC_BF16, true>), grid, block, 0, s, x_hi, lo, ldx, (const int *)nullptr, gamma, beta,
                           rows, width, eps, out16, ldo, out16_lo);
    else
        return ec::fail(EC_ERR_INVALID, "ec_layernorm_hl: unknown dtype %d", dtype);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

EC_API int ec_layernorm_hl8(const void *x_hi, const void *x_lo, long ldx, const float *gamma, const float *beta, int rows,
                            int width, float eps, void *out16, void *out_lo8, void *out_hi8, long ldo, int lo_exp, int hi_exp,
                            ec_stream_t stream)
{
    EC_REQUIRE(rows >= 0 && width > 0 && width % 4 == 0 && width <= LN_MAXV * 256,
               "ec_layernorm_hl8: width=%d must be a multiple of 4 and <= %d", width, LN_MAXV * 256);
    if (rows == 0) return EC_OK;
    EC_REQUIRE(x_hi && x_lo && gamma && beta && out16 && out_lo8, "ec_layernorm_hl8: null buffer");
    EC_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0 && ldo >= width, "ec_layernorm_hl8: strides must be multiples of 4 (ldo >= width)");
    EC_REQUIRE(lo_exp >= -60 && lo_exp <= 60 && hi_exp >= -60 && hi_exp <= 60, "ec_layernorm_hl8: exponents outside -60 .. 60");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid(ec::ceil_div(rows, 4)), block(256);
    ec::ProfScope prof(ec::PROF_LAYERNORM, s, 0, (double)rows * width * (out_hi8 ? 8.0 : 7.0));
    hipLaunchKernelGGL((layernorm_kernel<EC_F16, true, true>), grid, block, 0, s, x_hi, static_cast<const _Float16 *>(x_lo), ldx,
                       (const int *)nullptr, gamma, beta, rows, width, eps, out16, ldo, out_lo8, out_hi8, ldexpf(1.f, lo_exp),
                       ldexpf(1.f, hi_exp));
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

EC_API int ec_vit_embed(const float *patch, const float *cls, const float *pos, const float *gamma,
                        const float *beta, int n_img, int seq, int width, float eps, float *x,
                        ec_stream_t stream)
{
    return ec_vit_embed_train(patch, cls, pos, gamma, beta, n_img, seq, width, eps, x, nullptr, stream);
}

EC_API int ec_vit_embed_train(const float *patch, const float *cls, const float *pos, const float *gamma,
                              const float *beta, int n_img, int seq, int width, float eps, float *x,
                              float *pre, ec_stream_t stream)
{
    EC_REQUIRE(width % 4 == 0 && width <= LN_MAXV * 256 && seq >= 2, "ec_vit_embed: bad shape");
    if (n_img == 0) return EC_OK;
    const long rows = (long)n_img * seq;
    ec::ProfScope prof(ec::PROF_EMBED, static_cast<hipStream_t>(stream), 0, (double)rows * width * 8.0);
    hipLaunchKernelGGL(vit_embed_kernel<-1>, dim3((unsigned)ec::ceil_div(rows, 4L)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), patch, cls, pos, gamma, beta, n_img, seq,
                       width, eps, x, pre, (_Float16 *)nullptr);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

EC_API int ec_text_embed(const int32_t *tokens, const float *table, const float *pos, int n_txt,
                         int ctx, int width, int vocab, float *x, ec_stream_t stream)
{
    EC_REQUIRE(width % 4 == 0 && ctx > 0 && vocab > 0, "ec_text_embed: bad shape");
    if (n_txt == 0) return EC_OK;
    const long rows = (long)n_txt * ctx;
    ec::ProfScope prof(ec::PROF_EMBED, static_cast<hipStream_t>(stream), 0, (double)rows * width * 8.0);
    hipLaunchKernelGGL(text_embed_kernel, dim3((unsigned)ec::ceil_div(rows, 4L)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), tokens, table, pos, n_txt, ctx, width,
                       vocab, x);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

}  // extern "C"
