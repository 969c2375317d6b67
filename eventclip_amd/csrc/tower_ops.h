// Host-side pieces shared by the tower drivers (towers.hip: inference, vit_train.hip: fine-tuning):
// scratch carving, the ec_gemm call forms, the split-precision patch embedding.
#pragma once
#include "common.h"

namespace ec_tower {

constexpr float LN_EPS = 1e-5f;

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct Scratch {
    unsigned char *base;
    size_t off, cap;
    void *take(size_t bytes)
    {
        void *p = base ? base + off : nullptr;
        off += align_up(bytes);
        return p;
    }
};

// K-batch scratch of the low-latency mode (ec_vit_weights.low_latency): set by the tower driver for the duration of
// one call on the calling thread; NULL = every launch in a single pass
struct LatencyScratch {
    void *ws;
    size_t bytes;
};
inline LatencyScratch &latency_scratch()
{
    static thread_local LatencyScratch s = {nullptr, 0};
    return s;
}
constexpr size_t LATENCY_WS_BYTES = (size_t)80 << 20;   // 256 workgroups x one 256 x 256 fp32 tile, with slack

inline int gemm(int M, int N, int K, int dtype, int epi, const void *A, const void *W, const float *bias,
                void *C, ec_stream_t s, long ldc = 0, long lda = 0)
{
    ec_gemm_args g = {};
    g.M = M, g.N = N, g.K = K, g.dtype = dtype, g.epilogue = epi, g.variant = 0;
    g.A = A, g.lda = lda ? lda : K, g.W = W, g.bias = bias, g.C = C, g.ldc = ldc ? ldc : N;
    g.ws = latency_scratch().ws, g.ws_bytes = latency_scratch().bytes;
    return ec_gemm(&g, s);
}

// the folded-LayerNorm forms (EC_EPI_RESID_HL: aux = lo plane; EC_EPI_*_LN: row statistics + column sums)
// W_lo / A_lo (split-operand blocks): the weight's / the activation's lo part, one more MFMA product each in the same launch
inline int gemm_hl(int M, int N, int K, int dtype, const void *A, const void *W, const float *bias, void *x_hi,
                   void *x_lo, ec_stream_t s, long ldc = 0, float *row_sums = nullptr, const void *W_lo = nullptr,
                   const void *A_lo = nullptr)
{
    ec_gemm_args g = {};
    g.M = M, g.N = N, g.K = K, g.dtype = dtype, g.epilogue = EC_EPI_RESID_HL, g.variant = 0;
    g.A = A, g.lda = K, g.W = W, g.bias = bias, g.C = x_hi, g.ldc = ldc ? ldc : N, g.aux = x_lo;
    g.row_sums = row_sums, g.W_lo = W_lo, g.A_lo = A_lo;
    return ec_gemm(&g, s);
}
inline int gemm_ln(int M, int N, int K, int dtype, int epi, const void *A, const void *W, const float *bias,
                   const float *row_stats, long row_stats_stride, const float *col_sums, void *C, ec_stream_t s,
                   long ldc = 0, long lda = 0)
{
    ec_gemm_args g = {};
    g.M = M, g.N = N, g.K = K, g.dtype = dtype, g.epilogue = epi, g.variant = 0;
    g.A = A, g.lda = lda ? lda : K, g.W = W, g.bias = bias, g.C = C, g.ldc = ldc ? ldc : N;
    g.row_stats = row_stats, g.row_stats_stride = row_stats_stride, g.col_sums = col_sums;
    return ec_gemm(&g, s);
}

// 16-bit output from split operands (EC_EPI_STORE16 / EC_EPI_GELU16 with A_lo / W_lo; C_lo: the output's lo part too):
// the QKV and c_fc GEMMs of the split-operand blocks
inline int gemm_split16(int M, int N, int K, int dtype, int epi, const void *A, const void *A_lo, const void *W,
                        const void *W_lo, const float *bias, void *C, void *C_lo, ec_stream_t s)
{
    ec_gemm_args g = {};
    g.M = M, g.N = N, g.K = K, g.dtype = dtype, g.epilogue = epi, g.variant = 0;
    g.A = A, g.lda = K, g.W = W, g.bias = bias, g.C = C, g.ldc = N;
    g.A_lo = A_lo, g.W_lo = W_lo, g.aux = C_lo;
    return ec_gemm(&g, s);
}

// ... with the lo products on the FP8 matrix path (ec_vit_weights.lo_fp8): A_lo8 = the e4m3 lo part of A (ec_layernorm_hl8 /
// the e4m3 lo output of c_fc), W8 = the e4m3 copy of W; where the weight has a lo part: A8 (e4m3 copy of A) with W_lo8, or the
// 16-bit W_lo where no A8 exists (c_proj).  C_lo8: the output's lo part as e4m3 (GELU16; exponent LO8_EXP).
constexpr int LO8_EXP = 12;    // lo parts of activations as e4m3 of lo . 2^12: saturates where the activation exceeds 256
constexpr int HI8_EXP = 0;     // e4m3 copies of hi parts at scale 1: saturates beyond 448
struct Fp8Parts {
    const void *A_lo8, *W8, *A8, *W_lo8;
    int w8_exp, w_lo8_exp;
};
inline void fp8_args(ec_gemm_args &g, const Fp8Parts &f)
{
    g.A_lo8 = f.A_lo8, g.W8 = f.W8, g.a_lo8_exp = LO8_EXP, g.w8_exp = f.w8_exp;
    if (f.A8 && f.W_lo8) g.A8 = f.A8, g.W_lo8 = f.W_lo8, g.a8_exp = HI8_EXP, g.w_lo8_exp = f.w_lo8_exp;
}
inline int gemm_split16_f8(int M, int N, int K, int epi, const void *A, const void *W, const Fp8Parts &f, const float *bias, void *C,
                           void *C_lo, bool C_lo_e4m3, ec_stream_t s)
{
    ec_gemm_args g = {};
    g.M = M, g.N = N, g.K = K, g.dtype = EC_F16, g.epilogue = epi, g.variant = 0;
    g.A = A, g.lda = K, g.W = W, g.bias = bias, g.C = C, g.ldc = N;
    fp8_args(g, f);
    g.aux = C_lo;
    if (C_lo && C_lo_e4m3) g.aux_e4m3 = 1, g.aux_exp = LO8_EXP;
    return ec_gemm(&g, s);
}
inline int gemm_hl_f8(int M, int N, int K, const void *A, const void *W, const Fp8Parts &f, const void *W_lo, const float *bias,
                      void *x_hi, void *x_lo, ec_stream_t s, float *row_sums = nullptr)
{
    ec_gemm_args g = {};
    g.M = M, g.N = N, g.K = K, g.dtype = EC_F16, g.epilogue = EC_EPI_RESID_HL, g.variant = 0;
    g.A = A, g.lda = K, g.W = W, g.bias = bias, g.C = x_hi, g.ldc = N, g.aux = x_lo;
    g.row_sums = row_sums;
    fp8_args(g, f);
    g.W_lo = W_lo;
    return ec_gemm(&g, s);
}

// layernorm.hip: ln_pre'd embedding straight into the hi / lo planes; class rows of the planes back to fp32
int vit_embed_hl(const float *patch, const float *cls, const float *pos, const float *gamma, const float *beta,
                 int n_img, int seq, int width, float eps, void *x_hi, void *x_lo, int dtype, ec_stream_t stream);
int split_hl(const float *x, long n, void *x_hi, void *x_lo, int dtype, ec_stream_t stream);
int join_hl_rows(const void *x_hi, const void *x_lo, long ld, int rows, int width, float *out, int dtype,
                 ec_stream_t stream);

// attention.hip: 16-bit attention on a PLAIN q with every score scaled in fp32 (q rounded once; ec_attention's kernel
// multiplies q by the scale and rounds it again)
int attention_exact_scale(const void *qkv, void *out, int n_seq, int S, int width, int heads, int dtype, ec_stream_t stream);

#define EC_TRY(expr)                  \
    do {                              \
        int _rc = (expr);             \
        if (_rc != EC_OK) return _rc; \
    } while (0)

// Split-precision product: x.w = xl.wh + xh.wl + xh.wh accumulated in fp32 -- ONE launch since round 5 (ec_gemm_args.A_lo /
// W_lo: the three products run into the same accumulators; rounds 1 - 4 launched three GEMMs that read and wrote the
// fp32 C twice more).  w_lo == NULL: the weight IS its 16-bit value (ec_vit_weights.weights_exact16), the product with
// its lo part -- a sum of zeros -- is skipped, the same bits out.
inline int gemm3(int M, int N, int K, int dtype, bool accumulate, const void *a_hi, const void *a_lo,
                 const void *w_hi, const void *w_lo, const float *bias, float *C, ec_stream_t s)
{
    ec_gemm_args g = {};
    g.M = M, g.N = N, g.K = K, g.dtype = dtype, g.epilogue = accumulate ? EC_EPI_RESID32 : EC_EPI_STORE32, g.variant = 0;
    g.A = a_hi, g.lda = K, g.W = w_hi, g.bias = bias, g.C = C, g.ldc = N;
    g.A_lo = a_lo, g.W_lo = w_lo;
    return ec_gemm(&g, s);
}

// conv1 (kernel = stride = patch, no bias) as a GEMM over im2col rows, to fp32 accuracy: a patch
// row is [hi | lo | 0] (kpad wide) and conv_w = [w_hi | w_hi | 0], so the first launch gives
// x_hi.w_hi + x_lo.w_hi; the second adds x_hi.w_lo over the row's first klo columns (conv_w_lo =
// [w_lo | 0]: the lo values the row holds beyond 3 p^2 meet zeros).  0.6 % of the tower's flops; the
// rounding of pixels and conv1.weight to 16 bits would otherwise be ~8 % of the logit error budget
// (tools/rounding_budget.py).
inline int patch_embed(const ec_vit_weights *w, const void *patches, int rows, float *out, ec_stream_t s)
{
    const int klo = ((3 * w->patch * w->patch + 63) / 64) * 64;
    EC_TRY(gemm(rows, w->width, w->kpad, w->dtype, EC_EPI_STORE32, patches, w->conv_w, nullptr, out, s));
    if (!w->conv_w_lo) return EC_OK;   // conv1.weight IS its 16-bit value (weights_exact16): a sum of zeros skipped, the same bits
    return gemm(rows, w->width, klo, w->dtype, EC_EPI_RESID32, patches, w->conv_w_lo, nullptr, out, s, 0,
                w->kpad);
}

}  // namespace ec_tower
