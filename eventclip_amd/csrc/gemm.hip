// 16-bit MFMA GEMM with fused epilogues for the CLIP transformer blocks (gfx950).
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T + bias[N] )
//
// This one kernel family carries 96 % of the FLOPs of the reference's
// encode_image / encode_text (the nn.Linear /
// nn.MultiheadAttention in/out projections, c_fc, c_proj, conv1 as an im2col GEMM
// and the final projections of un-vendored openai/CLIP clip/model.py).
//
// Design (MI355X first):
//  * v_mfma_f32_16x16x32_{f16,bf16}, fp32 accumulate.  The weight tile is the
//    MFMA "A" operand and the activation tile the "B" operand, so a lane's four
//    accumulator registers are four consecutive output columns n of one row m;
//    the rows of the weight tile are additionally permuted so that each lane ends
//    up with 16 consecutive n: epilogue loads/stores are 16 B per lane and a
//    wave writes whole 128-B lines of a 16-bit output.
//  * Tiles are staged global -> LDS by 16-B LDS-DMA (global_load_lds_dwordx4),
//    two buffers.  LDS rows are 128 B (BK = 64); the 16-B chunk index is XORed
//    with key(row) = (row & 7) ^ ((row >> 3) & 6), applied on the DMA *source*
//    address (the DMA destination is lane-linear) and again on the ds_read_b128
//    side, which makes both the natural and the permuted fragment reads
//    bank-conflict free.
//  * blockIdx -> tile mapping is XCD-aware: the 8 XCDs each get a contiguous
//    range of tiles ordered N-fastest, so the N/BN tiles that share an
//    activation row panel hit the same L2.
#include <type_traits>

#include "common.h"
#include "mfma.h"

namespace {

using namespace ec;

constexpr int BK = 64;  // K elements per stage: 128-B LDS rows

struct GemmArgs {
    int M, N, K;
    const void *A;
    long lda;
    const void *W;
    const float *bias;
    void *C;
    long ldc;
    int tiles_m, tiles_n;
    unsigned long long *diag;   // EC_GEMM_DIAG builds: stamp / timeline records (ec_gemm_args.diag)
    // training extensions (persistent kernel only; see ec_gemm_args)
    long ldw;                   // row stride of W in elements
    const float *resid;         // RESID32: residual source (same ldc); null = C (in place)
    void *aux;                  // GELU16_SAVE: pre-activation out; GELU_BWD16: pre-activation in (ldc)
    int splits;                 // K-batches: batch s reads columns s*K .. of A and W, writes C + s * split_stride
    long split_stride;          // elements of C between batches
    // transposed operands (persistent kernel, STORE32): A is [rows, M] and W is [rows, N], the reduction runs over
    // their ROW index; K = rows per batch (a multiple of 64), k_valid = rows that exist (the rest read as zero)
    int tn;
    int k_valid;
    // LayerNorm folded into the GEMMs around it (EC_EPI_RESID_HL / EC_EPI_STORE16_LN / EC_EPI_GELU16_LN)
    const float *rowstat;       // consumers: [M][2] (rstd, -rstd * mean) of the A rows, row m at rowstat + 2 m rowstat_stride
    long rowstat_stride;
    const float *colsum;        // consumers: [N] sum over k of the (gamma-scaled, rounded) weight row
    float *stat_out;            // RESID_HL: optional [M][stat_groups][2] (sum, sum of squares) of the new hi plane per 64 columns
    int stat_groups;            // N / 64
    // Split-precision products in ONE launch (ec_gemm_args.A_lo / W_lo): the reduction runs over nseg segments of K
    // columns each, segment s multiplying the hi or the lo part of A by the hi or the lo part of W (the same row
    // strides), all into the same accumulators -- [a_lo | a_hi | a_hi] . [w_hi | w_lo | w_hi] without the operands being
    // laid out that way.
    int nseg;
    long seg_da, seg_dw;        // bytes from A / W (the hi parts) to the lo parts
    unsigned seg_mask;          // bit s: segment s reads A's lo part; bit 4 + s: W's lo part
    // FP8 lo products (SEG == 2, round 6): the FIRST nf8 (<= 2) segments multiply e4m3 operands -- 128 K elements per
    // 128-byte LDS row instead of 64, one v_mfma_scale_f32_16x16x128_f8f6f4 where the 16-bit tile issues two
    // v_mfma_f32_16x16x32 (the same matrix-pipe cycles per LDS tile, twice the K) -- stored at the SAME byte row pitch as the
    // 16-bit operands (the first K bytes of each row), so that the staging DMA's per-lane offsets are unchanged.
    int nf8;
    long f8_oa[2], f8_ow[2];    // bytes from A / W to the e4m3 operands of fp8 segment s
    int f8_scale[2];            // e8m0 byte (x 0x01010101) on the W side of segment s: 2^(byte - 127) undoes both operands' scales
    float aux8_scale;           // STORE16 / GELU16 with the lo output as e4m3 (HLM bit 3): lo . aux8_scale, one byte per element
};

// sixteen zero bytes for the LDS-DMA lanes whose reduction row does not exist (transposed operands)
__device__ __attribute__((aligned(16))) unsigned int tn_zero16[4];

// transposed-operand LDS image: [64 reduction rows][8 units of 32 B]; unit u of row r sits at u ^ tn_key(r), which
// spreads the 8 rows a 32-lane half of ds_read_b64_tr_b16 takes ({0..3, 8..11} + 4 hh + 16 n) over the 64 banks
__device__ __forceinline__ int tn_key(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }


__device__ __forceinline__ int swz_key(int row) { return (row & 7) ^ ((row >> 3) & 6); }

template <int CTRL> __device__ __forceinline__ float dpp_f32(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

__device__ __forceinline__ float quick_gelu(float x)
{
    // QuickGELU of openai/CLIP: x * sigmoid(1.702 x); v_exp_f32 + v_rcp_f32 (1 ulp each),
    // far inside the 16-bit rounding of the output.  The exponent's argument is ONE multiply (-1.702 log2 e folded
    // by hand: without -ffast-math hipcc keeps __expf(-1.702f * x) as two): the epilogue is bound by the vector
    // ALU's issue rate, 8.6 % of the c_fc GEMM went here
    return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * (-1.702f * 1.4426950408889634f)));
}
// four at a time: the multiply in front of the exponential and the add behind it as packed fp32 operations (two
// elements per 4-cycle issue; each still one correctly rounded fp32 operation, so the values are those of quick_gelu)
__device__ __forceinline__ f32x4 quick_gelu4(f32x4 x)
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x4 y;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const f32x2 v = {x[2 * h], x[2 * h + 1]};
        const f32x2 t = v * (-1.702f * 1.4426950408889634f);
        const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
        const f32x2 d = e + 1.f;
        const f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        const f32x2 o = v * r;
        y[2 * h] = o[0], y[2 * h + 1] = o[1];
    }
    return y;
}
// d QuickGELU / dx = s (1 + 1.702 x (1 - s)), s = sigmoid(1.702 x)
__device__ __forceinline__ float quick_gelu_grad(float x)
{
    const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * x));
    return sg * (1.f + 1.702f * x * (1.f - sg));
}
constexpr bool epi_is16(int e)
{
    return e == EC_EPI_STORE16 || e == EC_EPI_GELU16 || e == EC_EPI_GELU16_SAVE || e == EC_EPI_GELU_BWD16 ||
           e == EC_EPI_STORE16_LN || e == EC_EPI_GELU16_LN;
}
constexpr bool epi_is_ln(int e) { return e == EC_EPI_STORE16_LN || e == EC_EPI_GELU16_LN; }

// bijective XCD remap (blocks b and b+8 share an XCD): XCD x gets a contiguous id range
__device__ __forceinline__ int xcd_remap(int bid, int nblk)
{
    const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + (bid >> 3);
}

// (The direct epilogue, the general fp32 LDS-transposed epilogue and the general hi-lo epilogue are only used by the
// diagnostic kernels now: csrc/gemm_diag.inc.  The persistent kernel runs the buffer-addressed forms below; the
// general 16-bit form stays here for the training epilogues with a second output / input.)

// 16-bit counterpart: a 16 x 64 row group is 16 rows of 128 B (pitch 144 B); after the transpose
// 8 consecutive lanes cover one full 128-B row and a store instruction writes 8 whole lines.
// scratch: 2 x 16 x 144 bytes per wave.
template <int DT, int EPI, int TM>
__device__ __forceinline__ void epilogue16_lds(const GemmArgs &g, f32x4 (&acc)[TM][4], int m_base,
                                               int n_base, int lane, unsigned char *scratch,
                                               const float *lds_rowstat = nullptr)
{
    typedef typename T16<DT>::elem elem;
    typedef typename T16<DT>::v8 v8;
    static_assert(epi_is16(EPI), "16-bit outputs only");
    constexpr int PITCH = 144;
    const int q = lane >> 4, lr = lane & 15;
    f32x4 bias[4];
#pragma unroll
    for (int j = 0; j < 4; j++) bias[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nb = n_base + q * 16;
    if (g.bias && nb < g.N) {
#pragma unroll
        for (int j = 0; j < 4; j++) bias[j] = *reinterpret_cast<const f32x4 *>(g.bias + nb + 4 * j);
    }
    const int col = n_base + (lane & 7) * 8;   // this lane's 8 output columns after the transpose
    const bool col_ok = col < g.N;
    // LayerNorm folded in (EC_EPI_*_LN): A held the RAW rows x and W the gamma-scaled weight, so
    //   LN(x) . W^T + b = rstd (x . W'^T) - rstd mean colsum(W') + (b + W beta):
    // per lane one row (rstd, -rstd mean) per 16-row group and the column sums next to the (folded) bias
    f32x4 cs[4];
    float rs0[TM], rs1[TM];
    if constexpr (epi_is_ln(EPI)) {
#pragma unroll
        for (int j = 0; j < 4; j++)
            cs[j] = nb < g.N ? *reinterpret_cast<const f32x4 *>(g.colsum + nb + 4 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
        if (lds_rowstat) {
            // the persistent kernel had this wave row's 128 pairs brought into LDS by DMA while the main loop ran
            // (the statistics were written by another kernel a moment ago: a cold read, ~1 k cycles per tile if
            // it were issued here)
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const float2 r = *reinterpret_cast<const float2 *>(lds_rowstat + 2 * (i * 16 + lr));
                rs0[i] = r.x, rs1[i] = r.y;
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; i++) {
                int m = m_base + i * 16 + lr;
                m = m < g.M ? m : g.M - 1;
                const float2 r = *reinterpret_cast<const float2 *>(g.rowstat + 2 * (long)m * g.rowstat_stride);
                rs0[i] = r.x, rs1[i] = r.y;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < TM; i++) {
        unsigned char *buf = scratch + (i & 1) * 16 * PITCH;
        elem o[16];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            f32x4 v;
            if constexpr (epi_is_ln(EPI))
                v = acc[i][j] * rs0[i] + (cs[j] * rs1[i] + bias[j]);
            else
                v = acc[i][j] + bias[j];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float y = v[r];
                // (the multiplies and the add as packed instructions on four values at a time: measured, no gain)
                if constexpr (EPI == EC_EPI_GELU16 || EPI == EC_EPI_GELU16_SAVE || EPI == EC_EPI_GELU16_LN) y = quick_gelu(y);
                o[4 * j + r] = to16(y, elem());
            }
        }
        *reinterpret_cast<u32x4 *>(buf + lr * PITCH + q * 32) = *reinterpret_cast<const u32x4 *>(&o[0]);
        *reinterpret_cast<u32x4 *>(buf + lr * PITCH + q * 32 + 16) = *reinterpret_cast<const u32x4 *>(&o[8]);
        if constexpr (EPI == EC_EPI_GELU16_SAVE) {
            // the pre-activation goes out next to the activation (training keeps it for the backward pass):
            // second scratch buffer, same transpose
            unsigned char *buf2 = scratch + ((i & 1) ^ 1) * 16 * PITCH;
            elem u[16];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const f32x4 v = acc[i][j] + bias[j];
#pragma unroll
                for (int r = 0; r < 4; r++) u[4 * j + r] = to16(v[r], elem());
            }
            *reinterpret_cast<u32x4 *>(buf2 + lr * PITCH + q * 32) = *reinterpret_cast<const u32x4 *>(&u[0]);
            *reinterpret_cast<u32x4 *>(buf2 + lr * PITCH + q * 32 + 16) = *reinterpret_cast<const u32x4 *>(&u[8]);
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const int row = (lane >> 3) + 8 * p;
                const u32x4 v = *reinterpret_cast<const u32x4 *>(buf2 + row * PITCH + (lane & 7) * 16);
                const int m = m_base + i * 16 + row;
                if (m < g.M && col_ok)
                    *reinterpret_cast<u32x4 *>((elem *)g.aux + (long)m * g.ldc + col) = v;
            }
        }
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int row = (lane >> 3) + 8 * p;
            u32x4 v = *reinterpret_cast<const u32x4 *>(buf + row * PITCH + (lane & 7) * 16);
            const int m = m_base + i * 16 + row;
            if constexpr (EPI == EC_EPI_GELU_BWD16) {
                // out = dg * QuickGELU'(u), u = the saved pre-activation at the same [m, n]
                if (m < g.M && col_ok) {
                    const v8 uu = *reinterpret_cast<const v8 *>((const elem *)g.aux + (long)m * g.ldc + col);
                    v8 dg = __builtin_bit_cast(v8, v);
#pragma unroll
                    for (int e = 0; e < 8; e++)
                        dg[e] = to16((float)dg[e] * quick_gelu_grad((float)uu[e]), elem());
                    v = __builtin_bit_cast(u32x4, dg);
                }
            }
            if (m < g.M && col_ok)
                *reinterpret_cast<u32x4 *>((elem *)g.C + (long)m * g.ldc + col) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------
// Buffer-addressed forms of the three LDS-transposed epilogues (round 4): what the persistent kernel runs.
//  * A wave tile's global accesses go through a wave-uniform buffer descriptor of the tile (scalar registers) whose
//    range ends at the last row that exists, plus ONE per-lane 32-bit byte offset (parked out of range for the
//    lanes whose columns do not exist) plus a 32-bit add per row group: `buffer_* v, v_off, s[rsrc], 0 offen`.
//    The hardware's range check drops the stores and zeroes the loads of rows >= M and columns >= N, so edge
//    tiles and interior tiles run the SAME predicate-free code (a row's result cannot depend on which of the two
//    its tile is in this launch: batch invariance), and every store instruction is always issued, which is what
//    the counted hand-over wait of the tile loop needs.  (What is range-checked on gfx950 is the SUM of the vector
//    and the scalar offset -- measured in round 5, when a segment's part addressed through the scalar offset read
//    zeros: the row / column offsets here are vector offsets and the scalar one stays 0.)
//  * The general forms above spend 64 - 80 quarter-rate v_mul_lo_u32 / v_mad_u64_u32 and ~100 64-bit vector adds
//    per wave tile on `(long)m * ldc + col` -- more vector-ALU time than the arithmetic of the epilogue itself
//    (tile timelines: profiles/r4_gemm.md).  They stay for the diagnostic kernels and the training epilogues.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const void *origin, long bytes)
{
    // raw buffer (stride 0), dword3 = 0x00020000 (gfx9 / CDNA: 32-bit data format, no swizzle).  Every component goes
    // through readfirstlane: the values ARE wave-uniform, but where one of them was computed on the vector ALU (64-bit
    // products) a loop-carried descriptor lands in vector registers and every buffer instruction is wrapped in a
    // waterfall loop
    const unsigned long a = reinterpret_cast<unsigned long>(origin);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const int n = __builtin_amdgcn_readfirstlane((int)(bytes > 0 ? bytes : 0));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long)hi << 32) | lo), 0, n, 0x00020000);
}
__device__ __forceinline__ u32x4 bload16(__amdgpu_buffer_rsrc_t r, int voff)
{
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
}
__device__ __forceinline__ void bstore16(u32x4 v, __amdgpu_buffer_rsrc_t r, int voff)
{
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, 0, 0);
}
// a vector offset no tile's range reaches (a tile spans <= 256 rows x stride < 2^21 elements x 4 bytes = 2^31 is excluded by
// ec_gemm's checks: spans stay <= 2^30): loads give 0, stores are dropped.  0x40000000 and not INT_MAX: the row-group
// steps added to it (<= 31 x 16 x stride bytes < 2^30) must not overflow a signed int.
constexpr int BUF_OOB = 0x40000000;
// rows of a TMx16-row wave tile at m_base that exist, as a byte range of `pitch`-byte rows
__device__ __forceinline__ long tile_span(int M, int m_base, int rows, long pitch)
{
    const int left = M - m_base;
    return (long)(left < rows ? (left > 0 ? left : 0) : rows) * pitch;
}

// v_fma_mix_f32 reads fp16 halves of a packed register in place (no convert instruction): the arithmetic of the
// fp16 residual planes in 5.5 instructions per element instead of 9.5.  H = which half of the 32-bit register.
// Every form is ONE correctly rounded fp32 operation, the same value the separate convert + add / fma gives.
template <int H> __device__ __forceinline__ float mix_add16(unsigned a, unsigned b)     // (float)a.H + (float)b.H
{
    float d;
    if constexpr (H == 0) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b));
    else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
template <int H> __device__ __forceinline__ float mix_sub16(float x, unsigned h)        // x - (float)h.H
{
    float d;
    if constexpr (H == 0) asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(d) : "v"(x), "v"(h));
    else asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(d) : "v"(x), "v"(h));
    return d;
}
template <int H> __device__ __forceinline__ float mix_acc16(unsigned h, float c)        // c + (float)h.H
{
    float d;
    if constexpr (H == 0) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(c));
    else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(c));
    return d;
}
template <int H> __device__ __forceinline__ float mix_sq16(unsigned h, float c)         // fma(h.H, h.H, c)
{
    float d;
    if constexpr (H == 0) asm("v_fma_mix_f32 %0, %1, %1, %2 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(h), "v"(c));
    else asm("v_fma_mix_f32 %0, %1, %1, %2 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(h), "v"(c));
    return d;
}

// bias of this lane's 16 accumulator columns (zero where there is no bias / the columns do not exist)
__device__ __forceinline__ void load_bias(const GemmArgs &g, int nb, f32x4 (&bias)[4])
{
#pragma unroll
    for (int j = 0; j < 4; j++) bias[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (g.bias && nb < g.N) {
#pragma unroll
        for (int j = 0; j < 4; j++) bias[j] = *reinterpret_cast<const f32x4 *>(g.bias + nb + 4 * j);
    }
}

// (hi, lo) <- split(hi + lo + acc + bias), optional (sum, sum of squares) of the new hi values per 64 columns.
// The residual loads of row group i + DEPTH are issued BEFORE the stores of group i: vmcnt retires in issue order,
// so a load issued behind a store cannot deliver before that store is acknowledged -- with the stores first every
// row group waited for a write acknowledgement (30 k cycles per out_proj tile, 38 % of it).
// `between` runs once the epilogue's first loads (bias, the first residual row groups) are issued: the kernel puts the
// DMA requests of the next tile's first K tile there, so that they are YOUNGER than those loads -- the epilogue's first
// wait then is for its own data only (in-order vmcnt: waiting for a load also waits for everything issued before it).
// MODE (A / B switch, diagnostic build variants 30 .. 33): bit 0 = two scratch buffers, row group i + 1 written while
// the transposed reads of group i are in flight; bit 1 = the residual prefetch grows from DEPTH to the whole tile.
constexpr int HL_MODE_DEFAULT = 1;
template <int DT, int TM, int MODE, typename F>
__device__ __forceinline__ void epilogue_hl_buf(const GemmArgs &g, f32x4 (&acc)[TM][4], int m_base, int n_base,
                                                int lane, float *scratch, F &&between)
{
    constexpr bool PIPE = (MODE & 1) != 0, GROW = (MODE & 2) != 0;
    // cost-splitting forms of the diagnostic build (variants 34 .. 37, tools/bench_resid_split.py; WRONG results on purpose):
    // bit 5 no residual loads (the planes read as zero), bit 6 no store of the lo plane, bit 7 no store at all
    constexpr bool NOLOAD = (MODE & 32) != 0, NOLO = (MODE & 64) != 0, NOST = (MODE & 128) != 0;
    typedef typename T16<DT>::elem elem;
    typedef typename T16<DT>::v8 v8;
    constexpr int PITCH = 68;
    const int q = lane >> 4, lr = lane & 15;
    f32x4 bias[4];
    load_bias(g, n_base + q * 16, bias);
    const int c8 = lane & 7, r8 = lane >> 3;
    const long org = ((long)m_base * g.ldc + n_base) * 2;                  // wave-uniform byte offset of the tile
    const long span = tile_span(g.M, m_base, TM * 16, g.ldc * 2);
    const __amdgpu_buffer_rsrc_t rh = tile_rsrc(reinterpret_cast<const char *>(g.C) + org, span);
    const __amdgpu_buffer_rsrc_t rl = tile_rsrc(reinterpret_cast<const char *>(g.aux) + org, span);
    const bool col_ok = n_base + c8 * 8 < g.N;
    const int voff = col_ok ? (r8 * (int)g.ldc + c8 * 8) * 2 : BUF_OOB;    // this lane's bytes inside the tile
    const int step8 = (int)g.ldc * 16;                                     // bytes per 8 rows (scalar)
    const bool stats = g.stat_out != nullptr;
    float *st0 = g.stat_out + ((long)m_base * g.stat_groups + (n_base >> 6)) * 2;
    const int svoff = r8 * g.stat_groups * 2;                              // floats
    const int sstep8 = g.stat_groups * 16;
    const int rows_left = g.M - m_base - r8;                               // row 16 i + 8 p + r8 exists iff 16 i + 8 p < rows_left
    // Residual loads run DEPTH row groups ahead; the registers a finished row group frees (its 16 accumulators and
    // its 16 loaded values) take the loads of TWO later groups, so the whole tile's planes are in flight from the
    // second row group on (the epilogue is bound by the latency of these HBM reads, not by their bandwidth).
    constexpr int DEPTH = TM < 3 ? TM : 3;
    u32x4 xh[TM][2], xl[TM][2];
    auto fetch = [&](int i) {
#pragma unroll
        for (int p = 0; p < 2; p++) {
            if constexpr (NOLOAD) {
                xh[i][p] = xl[i][p] = u32x4{0u, 0u, 0u, 0u};
            } else {
                xh[i][p] = bload16(rh, voff + (2 * i + p) * step8);
                xl[i][p] = bload16(rl, voff + (2 * i + p) * step8);
            }
        }
    };
#pragma unroll
    for (int i = 0; i < DEPTH; i++) fetch(i);
    between();
    // two scratch buffers per wave: row group i + 1 is written while the transposed reads of group i are in flight
    auto produce = [&](int i) {
        float *buf = scratch + (PIPE ? (i & 1) * 16 * PITCH : 0);
#pragma unroll
        for (int j = 0; j < 4; j++)
            *reinterpret_cast<f32x4 *>(buf + lr * PITCH + q * 16 + j * 4) = acc[i][j] + bias[j];
    };
    if (PIPE) produce(0);
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const float *buf = scratch + (PIPE ? (i & 1) * 16 * PITCH : 0);
        if (!PIPE) produce(i);
        f32x4 ta[2], tb[2];
#pragma unroll
        for (int p = 0; p < 2; p++) {
            ta[p] = *reinterpret_cast<const f32x4 *>(buf + (r8 + 8 * p) * PITCH + c8 * 8);
            tb[p] = *reinterpret_cast<const f32x4 *>(buf + (r8 + 8 * p) * PITCH + c8 * 8 + 4);
        }
        if (PIPE && i + 1 < TM) produce(i + 1);
        u32x4 oh[2], ol[2];
        float ps[2], pq[2];
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const f32x4 a = ta[p], b = tb[p];
            ps[p] = 0.f, pq[p] = 0.f;
            if constexpr (DT == EC_F16) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const unsigned h2 = xh[i][p][k], l2 = xl[i][p][k];
                    const float a0 = k < 2 ? a[2 * k] : b[2 * k - 4], a1 = k < 2 ? a[2 * k + 1] : b[2 * k - 3];
                    const float x0 = mix_add16<0>(h2, l2) + a0, x1 = mix_add16<1>(h2, l2) + a1;
                    const f16x2 o = {(_Float16)x0, (_Float16)x1};
                    const unsigned o2 = __builtin_bit_cast(unsigned, o);
                    const f16x2 d = {(_Float16)mix_sub16<0>(x0, o2), (_Float16)mix_sub16<1>(x1, o2)};
                    oh[p][k] = o2, ol[p][k] = __builtin_bit_cast(unsigned, d);
                    ps[p] = mix_acc16<0>(o2, ps[p]), pq[p] = mix_sq16<0>(o2, pq[p]);
                    ps[p] = mix_acc16<1>(o2, ps[p]), pq[p] = mix_sq16<1>(o2, pq[p]);
                }
            } else {
                const v8 vh = __builtin_bit_cast(v8, xh[i][p]);
                const f16x8 vl = __builtin_bit_cast(f16x8, xl[i][p]);
                v8 wh;
                f16x8 wl;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float x = (float)vh[e] + (float)vl[e] + (e < 4 ? a[e] : b[e - 4]);
                    wh[e] = to16(x, elem());
                    const float h = (float)wh[e];
                    wl[e] = (_Float16)(x - h);
                    ps[p] += h, pq[p] = __builtin_fmaf(h, h, pq[p]);
                }
                oh[p] = __builtin_bit_cast(u32x4, wh), ol[p] = __builtin_bit_cast(u32x4, wl);
            }
        }
        // before the stores (see above); two groups per finished group until everything is requested
        if constexpr (GROW) {
            if (DEPTH + 2 * i < TM) fetch(DEPTH + 2 * i);
            if (DEPTH + 2 * i + 1 < TM) fetch(DEPTH + 2 * i + 1);
        } else {
            if (i + DEPTH < TM) fetch(i + DEPTH);
        }
#pragma unroll
        for (int p = 0; p < 2; p++) {
            if constexpr (!NOST) bstore16(oh[p], rh, voff + (2 * i + p) * step8);
            if constexpr (!NOST && !NOLO) bstore16(ol[p], rl, voff + (2 * i + p) * step8);
            if constexpr (NOST) asm volatile("" ::"v"(oh[p]), "v"(ol[p]));      // keep the arithmetic
            else if constexpr (NOLO) asm volatile("" ::"v"(ol[p]));
        }
        if (stats) {
#pragma unroll
            for (int p = 0; p < 2; p++) {
                float s = ps[p], qq = pq[p];
                s += dpp_f32<0xB1>(s), qq += dpp_f32<0xB1>(qq);      // quad_perm [1, 0, 3, 2]
                s += dpp_f32<0x4E>(s), qq += dpp_f32<0x4E>(qq);      // quad_perm [2, 3, 0, 1]
                s += dpp_f32<0x141>(s), qq += dpp_f32<0x141>(qq);    // row_half_mirror
                if (c8 == 0 && col_ok && 16 * i + 8 * p < rows_left)     // (a wave tile beyond N has no column group)
                    *reinterpret_cast<float2 *>(st0 + (2 * i + p) * sstep8 + svoff) = make_float2(s, qq);
            }
        }
    }
}

// 16-bit outputs (STORE16 / GELU16 and their folded-LayerNorm forms).  lds_rowstat: the wave row's 128 statistics
// pairs in LDS (has_lds; always a pointer INTO the shared array, so that the reads compile to ds_read and not to
// flat loads, whose s_waitcnt vmcnt(0) lgkmcnt(0) also waited for the next tile's first K tile), else g.rowstat
template <int DT, int EPI, int TM, bool HAS_LDS, bool EARLY = false, bool LOUT = false, bool LOUT8 = false, typename F>
__device__ __forceinline__ void epilogue16_buf(const GemmArgs &g, f32x4 (&acc)[TM][4], int m_base, int n_base,
                                               int lane, unsigned char *scratch, const float *lds_rowstat, F &&between)
{
    typedef typename T16<DT>::elem elem;
    static_assert(EPI == EC_EPI_STORE16 || EPI == EC_EPI_GELU16 || epi_is_ln(EPI), "inference epilogues only");
    constexpr int PITCH = 144;
    const int q = lane >> 4, lr = lane & 15;
    const int nb = n_base + q * 16;
    f32x4 bias[4];
    f32x4 cs[4];
    float rs0[TM], rs1[TM];
    if constexpr (epi_is_ln(EPI) && HAS_LDS) {
        // the statistics first: hipcc guards an LDS read behind LDS-DMA requests with a vmcnt wait (the pairs came by
        // DMA a tile ago), which is free here and would be a wait for the bias / column-sum loads behind them
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const float2 r = *reinterpret_cast<const float2 *>(lds_rowstat + 2 * (i * 16 + lr));
            rs0[i] = r.x, rs1[i] = r.y;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    load_bias(g, nb, bias);
    if constexpr (epi_is_ln(EPI)) {
#pragma unroll
        for (int j = 0; j < 4; j++)
            cs[j] = nb < g.N ? *reinterpret_cast<const f32x4 *>(g.colsum + nb + 4 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (HAS_LDS) {
        } else {
            // pairs [M][2] at a stride of rowstat_stride pairs (the class-token rows of the last block: a handful of
            // rows per launch), rows past M read the last pair.  (Plain global loads: hipcc 7.2 lowers
            // __builtin_amdgcn_raw_buffer_load_b64 to ONE dword here.)
#pragma unroll
            for (int i = 0; i < TM; i++) {
                int m = m_base + i * 16 + lr;
                m = m < g.M ? m : g.M - 1;
                const float2 r = *reinterpret_cast<const float2 *>(g.rowstat + 2 * (long)m * g.rowstat_stride);
                rs0[i] = r.x, rs1[i] = r.y;
            }
        }
    }
    if constexpr (EARLY) between();     // (A / B: the requests in front of the first use of bias, rounds 1 - 3 and early round 4)
    const __amdgpu_buffer_rsrc_t rc = tile_rsrc(reinterpret_cast<char *>(g.C) + ((long)m_base * g.ldc + n_base) * 2,
                                                tile_span(g.M, m_base, TM * 16, g.ldc * 2));
    const bool col_ok = n_base + (lane & 7) * 8 < g.N;
    const int voff = col_ok ? ((lane >> 3) * (int)g.ldc + (lane & 7) * 8) * 2 : BUF_OOB;
    const int step8 = (int)g.ldc * 16;
    // Row group i + 1 is converted and written to its scratch buffer while the transposed reads of row group i
    // (other buffer) are in flight: one LDS round trip per row group is hidden behind the next group's arithmetic
    auto produce = [&](int i) {
        unsigned char *buf = scratch + (i & 1) * 16 * PITCH;
        elem o[16];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            f32x4 v;
            if constexpr (epi_is_ln(EPI))
                v = acc[i][j] * rs0[i] + (cs[j] * rs1[i] + bias[j]);
            else
                v = acc[i][j] + bias[j];
            if constexpr (EPI == EC_EPI_GELU16 || EPI == EC_EPI_GELU16_LN) v = quick_gelu4(v);
#pragma unroll
            for (int r = 0; r < 4; r++) o[4 * j + r] = to16(v[r], elem());
        }
        *reinterpret_cast<u32x4 *>(buf + lr * PITCH + q * 32) = *reinterpret_cast<const u32x4 *>(&o[0]);
        *reinterpret_cast<u32x4 *>(buf + lr * PITCH + q * 32 + 16) = *reinterpret_cast<const u32x4 *>(&o[8]);
    };
    if constexpr (LOUT) {
        // Split output (ec_gemm_args.aux with an *_LN epilogue): C = hi = round16(v), aux = lo = round16(v - hi), the
        // operand pair of a split-precision consumer (ec_gemm_args.A_lo; the attention of the split-operand blocks).
        // One scratch buffer per part, no pipelining of the row groups (this form runs in a few blocks of the tolerance
        // mode only); twice the stores.
        const __amdgpu_buffer_rsrc_t rl = tile_rsrc(reinterpret_cast<char *>(g.aux) + ((long)m_base * g.ldc + n_base) * 2,
                                                    tile_span(g.M, m_base, TM * 16, g.ldc * 2));
        // e4m3 lo output: row m of aux starts at byte m * 2 ldc (the 16-bit pitch), column n is byte n
        const __amdgpu_buffer_rsrc_t rl8 = tile_rsrc(reinterpret_cast<char *>(g.aux) + (long)m_base * g.ldc * 2 + n_base,
                                                     tile_span(g.M, m_base, TM * 16, g.ldc * 2));
        const int voff8 = col_ok ? (lane >> 3) * (int)g.ldc * 2 + (lane & 7) * 8 : BUF_OOB;
#pragma unroll
        for (int i = 0; i < TM; i++) {
            elem o[16], l[16];
            float l32[16];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                f32x4 v;
                if constexpr (epi_is_ln(EPI))
                    v = acc[i][j] * rs0[i] + (cs[j] * rs1[i] + bias[j]);
                else
                    v = acc[i][j] + bias[j];
                if constexpr (EPI == EC_EPI_GELU16 || EPI == EC_EPI_GELU16_LN) v = quick_gelu4(v);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float x = v[r];
                    asm volatile("" : "+v"(x));      // ONE rounded fp32 value for both parts (see attention_f32m_kernel)
                    o[4 * j + r] = to16(x, elem());
                    l32[4 * j + r] = x - (float)o[4 * j + r];
                    l[4 * j + r] = to16(l32[4 * j + r], elem());
                }
            }
            unsigned char *b0 = scratch, *b1 = scratch + 16 * PITCH;
            *reinterpret_cast<u32x4 *>(b0 + lr * PITCH + q * 32) = *reinterpret_cast<const u32x4 *>(&o[0]);
            *reinterpret_cast<u32x4 *>(b0 + lr * PITCH + q * 32 + 16) = *reinterpret_cast<const u32x4 *>(&o[8]);
            if constexpr (LOUT8) {
                // the lo part as e4m3 of lo . aux8_scale (ec_gemm_args.aux_e4m3): 16 bytes per lane, a row of the 16 x 64 group =
                // 64 bytes; aux rows lie at the byte pitch of C (the A_lo8 operand of the GEMM that follows)
                u32x4 l8;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float d[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) d[r] = __builtin_amdgcn_fmed3f(l32[4 * j + r] * g.aux8_scale, -448.f, 448.f);
                    unsigned w = __builtin_amdgcn_cvt_pk_fp8_f32(d[0], d[1], 0u, false);
                    l8[j] = __builtin_amdgcn_cvt_pk_fp8_f32(d[2], d[3], w, true);
                }
                *reinterpret_cast<u32x4 *>(b1 + lr * PITCH + q * 16) = l8;
            } else {
                *reinterpret_cast<u32x4 *>(b1 + lr * PITCH + q * 32) = *reinterpret_cast<const u32x4 *>(&l[0]);
                *reinterpret_cast<u32x4 *>(b1 + lr * PITCH + q * 32 + 16) = *reinterpret_cast<const u32x4 *>(&l[8]);
            }
            if (i == 0) {
                __builtin_amdgcn_sched_barrier(0);
                between();
                __builtin_amdgcn_sched_barrier(0);
            }
            u32x4 t[2], u[2];
            u32x2 u8[2];
#pragma unroll
            for (int p = 0; p < 2; p++) {
                t[p] = *reinterpret_cast<const u32x4 *>(b0 + ((lane >> 3) + 8 * p) * PITCH + (lane & 7) * 16);
                if constexpr (LOUT8) u8[p] = *reinterpret_cast<const u32x2 *>(b1 + ((lane >> 3) + 8 * p) * PITCH + (lane & 7) * 8);
                else u[p] = *reinterpret_cast<const u32x4 *>(b1 + ((lane >> 3) + 8 * p) * PITCH + (lane & 7) * 16);
            }
#pragma unroll
            for (int p = 0; p < 2; p++) {
                bstore16(t[p], rc, voff + (2 * i + p) * step8);
                if constexpr (LOUT8) __builtin_amdgcn_raw_buffer_store_b64(u8[p], rl8, voff8 + (2 * i + p) * step8, 0, 0);
                else bstore16(u[p], rl, voff + (2 * i + p) * step8);
            }
        }
        return;
    }
    produce(0);
    // The next tile's DMA is requested HERE, behind the first use of bias / column sums / row statistics: hipcc's
    // wait-count pass does not count the LDS-DMA requests when it waits for those loads (it asks for vmcnt(1) and
    // vmcnt(0) where 9 and 8 would do), so with the requests in front of that use every tile's epilogue opened with a
    // wait for the whole first K tile of the next one
    if constexpr (!EARLY) {
        __builtin_amdgcn_sched_barrier(0);     // (all of produce(0), hence every such wait, stays in front of the requests)
        between();
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const unsigned char *buf = scratch + (i & 1) * 16 * PITCH;
        u32x4 t[2];
#pragma unroll
        for (int p = 0; p < 2; p++)
            t[p] = *reinterpret_cast<const u32x4 *>(buf + ((lane >> 3) + 8 * p) * PITCH + (lane & 7) * 16);
        if (i + 1 < TM) produce(i + 1);
#pragma unroll
        for (int p = 0; p < 2; p++) bstore16(t[p], rc, voff + (2 * i + p) * step8);
    }
}

// fp32 outputs (STORE32 / RESID32); residual loads ahead of the stores as in epilogue_hl_buf
template <int EPI, int TM, bool NAT, typename F>
__device__ __forceinline__ void epilogue32_buf(const GemmArgs &g, f32x4 (&acc)[TM][4], int m_base, int n_base,
                                               int lane, float *scratch, F &&between)
{
    static_assert(EPI == EC_EPI_RESID32 || EPI == EC_EPI_STORE32, "fp32 outputs only");
    constexpr int PITCH = 68;
    const int q = lane >> 4, lr = lane & 15;
    f32x4 bias[4];
    load_bias(g, n_base + q * 16, bias);
    const long org = ((long)m_base * g.ldc + n_base) * 4, span = tile_span(g.M, m_base, TM * 16, g.ldc * 4);
    const __amdgpu_buffer_rsrc_t rc = tile_rsrc(reinterpret_cast<char *>(g.C) + org, span);
    const __amdgpu_buffer_rsrc_t rr = g.resid ? tile_rsrc(reinterpret_cast<const char *>(g.resid) + org, span) : rc;
    const bool col_ok = n_base + lr * 4 < g.N;
    const int voff = col_ok ? (q * (int)g.ldc + lr * 4) * 4 : BUF_OOB;
    const int step4 = (int)g.ldc * 16;                                     // bytes per 4 rows
    constexpr int DEPTH = EPI == EC_EPI_RESID32 ? (TM < 3 ? TM : 3) : 1;
    f32x4 x[DEPTH][4];
    auto fetch = [&](int i) {
        if constexpr (EPI == EC_EPI_RESID32) {
#pragma unroll
            for (int p = 0; p < 4; p++) x[i % DEPTH][p] = __builtin_bit_cast(f32x4, bload16(rr, voff + (4 * i + p) * step4));
        }
    };
    if constexpr (EPI == EC_EPI_RESID32) {
#pragma unroll
        for (int i = 0; i < DEPTH; i++) fetch(i);
    }
    between();
#pragma unroll
    for (int i = 0; i < TM; i++) {
        float *buf = scratch;
#pragma unroll
        for (int j = 0; j < 4; j++)
            *reinterpret_cast<f32x4 *>(buf + lr * PITCH + (NAT ? j * 16 + q * 4 : q * 16 + j * 4)) = acc[i][j] + bias[j];
        f32x4 v[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            v[p] = *reinterpret_cast<const f32x4 *>(buf + (q + 4 * p) * PITCH + lr * 4);
            if constexpr (EPI == EC_EPI_RESID32) v[p] += x[i % DEPTH][p];
        }
        if constexpr (EPI == EC_EPI_RESID32) {
            if (i + DEPTH < TM) fetch(i + DEPTH);
        }
#pragma unroll
        for (int p = 0; p < 4; p++) bstore16(__builtin_bit_cast(u32x4, v[p]), rc, voff + (4 * i + p) * step4);
    }
}

#define EC_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// (diagnostic build) shader-clock and 100 MHz reference counters of this workgroup -> diag[4 b + at], diag[4 b + at + 1]
__device__ __forceinline__ void clock_stamp(unsigned long long *diag, int at)
{
    unsigned long long c, r;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c), "=s"(r)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    if (threadIdx.x == 0) diag[4 * blockIdx.x + at] = c, diag[4 * blockIdx.x + at + 1] = r;
}

// Tile raster inside an XCD's contiguous id range: 8 row panels x 4 column tiles per group of
// 32 ids (one per CU of the XCD), so the 32 workgroups an XCD runs at a time stream 8 + 4
// distinct operand panels through its L2 instead of 2-3 + tiles_n.  Rows beyond the last full
// group of 8 (and tilings whose column count is not a multiple of 4) keep the N-fastest order.
__device__ __forceinline__ void raster(int id, int tiles_m, int tiles_n, int &tm, int &tn)
{
    const int full = (tiles_n & 3) == 0 ? (tiles_m >> 3) * 8 * tiles_n : 0;
    if (id < full) {
        const int per_group = 8 * tiles_n;
        const int grp = id / per_group, r = id - grp * per_group;
        const int chunk = r >> 5, w = r & 31;
        tm = grp * 8 + (w >> 2);
        tn = chunk * 4 + (w & 3);
    } else {
        const int r = id - full, base_m = full / tiles_n;
        tm = base_m + r / tiles_n;
        tn = r % tiles_n;
    }
}


// ---------------------------------------------------------------------------------------
// Persistent form of the staggered two-phase kernel: one workgroup per CU walks tiles
// blockIdx.x, blockIdx.x + gridDim.x, ...  (the same XCD / raster order as the one-tile-per-
// workgroup launch, 256 tiles in flight at a time).  The first K tile of the next output tile
// is on its way into staging buffer 0 while the epilogue drains the accumulators through a
// scratch area in buffer 1, so neither the workgroup turnaround (~1.5 k cycles) nor the
// prologue's HBM latency (~2.9 k cycles of a 48 k-cycle tile at K = 1024) is exposed.
// TL (diagnostic build): 1 = per-tile timeline records as in gemm2p_kernel<DBG = 9>; 2 = nothing but (s_memtime,
// s_memrealtime) of wave 0 at the workgroup's start and end -> args.diag[4 b .. 4 b + 3]: the in-kernel clock
// (shader cycles per 100 MHz tick) with no stamp inside the tile loop.
// ---------------------------------------------------------------------------------------
// SEG: the reduction runs over g.nseg segments (split-precision operands, see GemmArgs): K tile t belongs to segment
// t / (K / 64).  A segment's parts are reached by rebuilding the tile's descriptors on the other part at the segment
// change (2 - 3 times per tile, scalar work; the instruction's scalar offset is no way there: it IS part of the range
// check on gfx950 -- offset + soffset >= num_records reads zeros -- so a range that covers it no longer ends at the
// tile's last row).  Region 1 is requested one K tile apart from regions 0, 2, 3 and has a descriptor of its own.
template <int DT, int EPI, int TL = 0, bool TN = false, int HLM = HL_MODE_DEFAULT, int SEG = 0>
__global__ __launch_bounds__(512) void gemm2pp_kernel(const GemmArgs g)
{
    typedef typename T16<DT>::v8 v8;
    typedef typename T16<DT>::v4 v4;
    static_assert(!TN || EPI == EC_EPI_STORE32, "transposed operands: fp32 store only");
    constexpr int BM = 256, BN = 256;
    constexpr int REGION = 128 * 128;
    constexpr int KT = 4 * REGION;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int ntiles_mn = g.tiles_m * g.tiles_n;
    const int ntiles = ntiles_mn * g.splits;   // K-batches are further tiles of the same launch
    static_assert(!SEG || !TN, "segments: row-major operands only");
    static_assert(SEG != 2 || DT == EC_F16, "fp8 lo products go with f16 hi products");
    // SEG == 2: the first g.nf8 segments are e4m3 (K / 128 tiles each), the rest 16-bit (K / 64 tiles each)
    const int nk8 = SEG == 2 ? g.nf8 * (g.K / (2 * BK)) : 0;
    const int nk = SEG == 2 ? nk8 + (g.nseg - g.nf8) * (g.K / BK) : SEG ? g.nseg * (g.K / BK) : g.K / BK;

    auto key = [](int row) { return (row & 7) ^ (((row >> 4) & 1) << 2); };

    int m0 = 0, n0 = 0, sp0 = 0;
    const unsigned char *src[4][2];     // transposed operands only (the row-major form addresses through descriptors)
    // Row-major operands (round 4): the staging DMA is buffer-addressed.  Per tile two descriptors (scalar registers:
    // the tile's A rows / W rows from this batch's first column, ranges ending at the last row that exists -- rows
    // past M or N read as zeros), per lane ONE 32-bit byte offset per operand that never changes (row lane >> 3 of a
    // piece, swizzled 16-byte chunk), per piece a scalar row offset, per region the bytes advanced along K: a request
    // costs one 32-bit add.  The pointer form kept sixteen 64-bit pointers (32 registers of the main loop's 249),
    // advanced each with a 64-bit add per K tile and rebuilt with 48 quarter-rate multiplies per tile; and as a
    // FLAT-encoded instruction that also touches LDS, global_load_lds makes hipcc's wait-count pass treat vmcnt as
    // unordered (every wait of its own behind one becomes vmcnt(0)).
    __amdgpu_buffer_rsrc_t rsA = tile_rsrc(g.A, 0), rsW = tile_rsrc(g.W, 0);
    int vbA = 0, vbW = 0, srow[4][2] = {}, kk[4] = {0, 0, 0, 0};
    if constexpr (!TN) {
        const int l3 = lane >> 3;
        // key(rr) of piece row rr = (8 i + wave) 8 + l3:  rr & 7 = l3,  bit 4 of rr = bit 1 of wave
        const int chunk = (lane & 7) ^ l3 ^ (((wave >> 1) & 1) << 2);
        vbA = (l3 * (int)g.lda + chunk * 8) * 2;
        vbW = (l3 * (int)g.ldw + chunk * 8) * 2;
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 2; i++)
                srow[r][i] = r < 2 ? (i * 128 + r * 64 + wave * 8) * (int)g.lda * 2
                                   : ((i * 2 + (wave >> 2)) * 64 + (wave & 3) * 16 + (r - 2) * 8) * (int)g.ldw * 2;
    }
    // transposed operands: reduction rows of this batch still to be issued per region (rows past k_valid read
    // tn_zero16), and the bytes one K tile advances an operand's pointer by
    // SEG: regions 0, 2, 3 are always requested together (stream X, advanced behind region 3), region 1 a K tile apart
    // (stream Y): bytes into the segment and the segment's index, per stream
    int kx = 0, ky = 0, sx = 0, sy = 0;
    const int kseg = g.K * 2;
    __amdgpu_buffer_rsrc_t rsAy = rsA;
    auto seg_rsrc = [&](int sg, bool w) {
        long d = ((g.seg_mask >> (w ? 4 + sg : sg)) & 1u) ? (w ? g.seg_dw : g.seg_da) : 0;
        if constexpr (SEG == 2) {
            if (sg < g.nf8) d = w ? (sg == 0 ? g.f8_ow[0] : g.f8_ow[1]) : (sg == 0 ? g.f8_oa[0] : g.f8_oa[1]);
        }
        return w ? tile_rsrc(static_cast<const unsigned char *>(g.W) + (long)n0 * g.ldw * 2 + d, tile_span(g.N, n0, BN, g.ldw * 2))
                 : tile_rsrc(static_cast<const unsigned char *>(g.A) + (long)m0 * g.lda * 2 + d, tile_span(g.M, m0, BM, g.lda * 2));
    };
    int rem[4] = {0, 0, 0, 0};
    const long adv_a = TN ? (long)g.lda * BK * 2 : BK * 2, adv_w = TN ? (long)g.ldw * BK * 2 : BK * 2;
    auto setup = [&](int id) {
        int tm, tn;
        // lane id re-derived per tile, opaque to the optimiser: the per-lane row / chunk constants below are cheap to
        // recompute and must not live (= spill) across the main loop; a scratch reload sits behind s_waitcnt vmcnt
        int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(lane));
        int rid = xcd_remap(id, ntiles);
        sp0 = 0;
        if (g.splits > 1) {
            sp0 = rid / ntiles_mn;
            rid -= sp0 * ntiles_mn;
        }
        const long koff = (long)sp0 * g.K;       // this batch's first column of A and W
        raster(rid, g.tiles_m, g.tiles_n, tm, tn);
        m0 = tm * BM, n0 = tn * BN;
        if constexpr (TN) {
            // LDS row = reduction row (4 per 1-KiB DMA piece), 16-byte slot = lane & 15 = 2 (unit ^ key) + half;
            // unit = 4 wm + mt for the A regions (columns m0 + 128 wm + 64 mq + 16 mt ..), 2 wn + jj for the W
            // regions (columns n0 + 64 wn + 32 nq + 16 jj ..)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                rem[r] = g.k_valid - (int)koff;
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const int krow = (i * 8 + wave) * 4 + (lane >> 4);
                    const int unit = ((lane & 15) >> 1) ^ tn_key(krow), h8 = (lane & 1) * 8;
                    if (r < 2) {
                        int col = m0 + (unit >> 2) * 128 + r * 64 + (unit & 3) * 16 + h8;
                        col = col + 8 <= g.M ? col : g.M - 8;
                        src[r][i] = (const unsigned char *)g.A + ((koff + krow) * g.lda + col) * 2;
                    } else {
                        int col = n0 + (unit >> 1) * 64 + (r - 2) * 32 + (unit & 1) * 16 + h8;
                        col = col + 8 <= g.N ? col : g.N - 8;
                        src[r][i] = (const unsigned char *)g.W + ((koff + krow) * g.ldw + col) * 2;
                    }
                }
            }
            return;
        }
        rsA = tile_rsrc(static_cast<const unsigned char *>(g.A) + ((long)m0 * g.lda + koff) * 2, tile_span(g.M, m0, BM, g.lda * 2));
        rsW = tile_rsrc(static_cast<const unsigned char *>(g.W) + ((long)n0 * g.ldw + koff) * 2, tile_span(g.N, n0, BN, g.ldw * 2));
#pragma unroll
        for (int r = 0; r < 4; r++) kk[r] = 0;
        kx = ky = sx = sy = 0;
        if constexpr (SEG) rsA = rsAy = seg_rsrc(0, false), rsW = seg_rsrc(0, true);
    };
    auto issue = [&](int r, int buf) {
        if (TN && rem[r] < BK) {       // the batch's last K tile runs past the rows that exist
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int krow = (i * 8 + wave) * 4 + (lane >> 4);
                const unsigned char *p = krow < rem[r] ? src[r][i] : reinterpret_cast<const unsigned char *>(tn_zero16);
                glds16(p, smem + buf * KT + r * REGION + (i * 8 + wave) * 1024);
            }
        } else if constexpr (TN) {
#pragma unroll
            for (int i = 0; i < 2; i++) glds16(src[r][i], smem + buf * KT + r * REGION + (i * 8 + wave) * 1024);
        } else if constexpr (SEG) {
            int &kb = r == 1 ? ky : kx, &sg = r == 1 ? sy : sx;
            int vb = r < 2 ? vbA : vbW;
            // SEG == 2: the per-lane base stays ONE register per operand (opaque here): hoisted as eight `base + row offset`
            // sums they did not fit beside the two loops' fragments and came back from scratch behind s_waitcnt vmcnt(0) at
            // every tile hand-over
            if constexpr (SEG == 2) asm volatile("" : "+v"(vb));
#pragma unroll
            for (int i = 0; i < 2; i++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r == 1 ? rsAy : r == 0 ? rsA : rsW,
                                                         (__attribute__((address_space(3))) void *)(smem + buf * KT + r * REGION + (i * 8 + wave) * 1024),
                                                         16, vb + (srow[r][i] + kb), 0, 0, 0);
            if (r == 1 || r == 3) {
                kb += BK * 2;
                if (kb == ((SEG == 2 && sg < g.nf8) ? (kseg >> 1) : kseg)) {
                    kb = 0, sg++;
                    if (sg < g.nseg) {
                        if (r == 1) rsAy = seg_rsrc(sg, false);
                        else rsA = seg_rsrc(sg, false), rsW = seg_rsrc(sg, true);
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; i++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r < 2 ? rsA : rsW,
                                                         (__attribute__((address_space(3))) void *)(smem + buf * KT + r * REGION + (i * 8 + wave) * 1024),
                                                         16, (r < 2 ? vbA : vbW) + (srow[r][i] + kk[r]), 0, 0, 0);
            kk[r] += BK * 2;
        }
        if constexpr (TN) {
#pragma unroll
            for (int i = 0; i < 2; i++) src[r][i] += r < 2 ? adv_a : adv_w;
            rem[r] -= BK;
        }
    };

    int offM[4], offN[2];
#pragma unroll
    for (int mt = 0; mt < 4; mt++) {
        const int row = wm * 64 + mt * 16 + (lane & 15);
        offM[mt] = row * 128 + (((lane >> 4) ^ key(row)) << 4);
    }
#pragma unroll
    for (int jj = 0; jj < 2; jj++) {
        const int i = lane & 15;
        const int row = wn * 32 + ((i >> 2) << 3) + (jj << 2) + (i & 3);
        offN[jj] = row * 128 + (((lane >> 4) ^ key(row)) << 4);
    }

    if constexpr (TN) {
        // transposed reads: lane 4 q + p of a 16-lane group addresses row 8 g + 4 hh + q (+ 32 ks), 8-byte piece p
        // of the fragment's 32-byte unit; the key is the same for every row this lane addresses
        const int q = (lane & 15) >> 2, p = lane & 3, gg = lane >> 4;
        const int key = q | ((gg & 1) << 2);
        const int base = (8 * gg + q) * 256 + ((p >> 1) << 4) + ((p & 1) << 3);
#pragma unroll
        for (int mt = 0; mt < 4; mt++) offM[mt] = base + (((wm * 4 + mt) ^ key) << 5);
#pragma unroll
        for (int jj = 0; jj < 2; jj++) offN[jj] = base + (((wn * 2 + jj) ^ key) << 5);
    }
    auto tr8 = [&](const unsigned char *at) {
        v8 f;
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(at + hh * 1024));
            const v4 tv = __builtin_bit_cast(v4, t);
            f[4 * hh] = tv[0], f[4 * hh + 1] = tv[1], f[4 * hh + 2] = tv[2], f[4 * hh + 3] = tv[3];
        }
        return f;
    };

    f32x4 acc[8][4];
    v8 fm[4][2], fn0[2][2], fn1[2][2];
    // SEG == 2: a lane's two 16-byte pieces of a row live in ONE 8-register tuple (the operand of the e4m3 product as it
    // stands; the 16-bit products read its halves) -- with separate 4-register values the allocator copied every operand of
    // the e4m3 products into a fresh tuple and spilled
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef int i32x8 __attribute__((ext_vector_type(8)));
    i32x8 gm[4], gn0[2], gn1[2];
    auto ld8 = [&](int base, int off) {      // (LDS-typed addresses: an XOR on a generic pointer compiled to flat loads)
        const i32x4 a = *reinterpret_cast<const i32x4 *>(smem + base + off);
        const i32x4 b = *reinterpret_cast<const i32x4 *>(smem + base + (off ^ 64));
        return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto load_m8 = [&](int buf, int mq) {
#pragma unroll
        for (int mt = 0; mt < 4; mt++) gm[mt] = ld8(buf * KT + mq * REGION, offM[mt]);
    };
    auto load_m = [&](int buf, int mq) {
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                if constexpr (TN)
                    fm[mt][ks] = tr8(smem + buf * KT + mq * REGION + offM[mt] + ks * 8192);
                else
                    fm[mt][ks] = *reinterpret_cast<const v8 *>(smem + buf * KT + mq * REGION +
                                                               (offM[mt] ^ (ks << 6)));
            }
    };
    auto load_n8 = [&](i32x8(&gn)[2], int buf, int nq) {
#pragma unroll
        for (int jj = 0; jj < 2; jj++) gn[jj] = ld8(buf * KT + (2 + nq) * REGION, offN[jj]);
    };
    auto load_n = [&](v8(&fn)[2][2], int buf, int nq) {
#pragma unroll
        for (int jj = 0; jj < 2; jj++)
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                if constexpr (TN)
                    fn[jj][ks] = tr8(smem + buf * KT + (2 + nq) * REGION + offN[jj] + ks * 8192);
                else
                    fn[jj][ks] = *reinterpret_cast<const v8 *>(smem + buf * KT + (2 + nq) * REGION +
                                                               (offN[jj] ^ (ks << 6)));
            }
    };
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mma2 = [&](int mq, int nqa, v8(&fa)[2][2], int nqb, v8(&fb)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
#pragma unroll
                for (int jj = 0; jj < 2; jj++)
                    acc[mq * 4 + mt][nqa * 2 + jj] =
                        mfma16(fa[jj][ks], fm[mt][ks], acc[mq * 4 + mt][nqa * 2 + jj]);
#pragma unroll
                for (int jj = 0; jj < 2; jj++)
                    acc[mq * 4 + mt][nqb * 2 + jj] =
                        mfma16(fb[jj][ks], fm[mt][ks], acc[mq * 4 + mt][nqb * 2 + jj]);
            }
        __builtin_amdgcn_s_setprio(0);
    };
    // e4m3 tile: the SAME fragment reads (a lane's two 16-byte pieces of a row, chunks g and g + 4 of the 128-byte row) are
    // the lane's 32 bytes of ONE 16x16x128 product -- which 32 of the 128 K elements a lane group holds does not matter as
    // long as both operands hold the same ones, and both are read through the same layout.  scale_w: e8m0 x 4 (uniform).
    auto mma2_g = [&](int mq, int nqa, i32x8(&ga)[2], int nqb, i32x8(&gb)[2], int scale_w) {
        if constexpr (SEG == 2) {
            // operand format code of the scaled product: 0 = e4m3.  -DEC_LO_FMT=2 (e2m3, FP6) / 4 (e2m1, FP4) in a diagnostic
            // build reads the SAME bytes as that format -- a rate experiment with meaningless results (tools/build_lo_fmt_probe.sh)
#if defined(EC_GEMM_DIAG) && defined(EC_LO_FMT)
            constexpr int FMT = EC_LO_FMT;
#else
            constexpr int FMT = 0;
#endif
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
#pragma unroll
                for (int jj = 0; jj < 2; jj++)
                    acc[mq * 4 + mt][nqa * 2 + jj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                        ga[jj], gm[mt], acc[mq * 4 + mt][nqa * 2 + jj], FMT, FMT, 0, scale_w, 0, 0x7f7f7f7f);
#pragma unroll
                for (int jj = 0; jj < 2; jj++)
                    acc[mq * 4 + mt][nqb * 2 + jj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                        gb[jj], gm[mt], acc[mq * 4 + mt][nqb * 2 + jj], FMT, FMT, 0, scale_w, 0, 0x7f7f7f7f);
            }
            // The products are pinned HERE: without it hipcc sank the sixteen MFMAs of phase A out of their slot between the two
            // barriers (sched_barrier does not stop the IR-level code motion of a pure intrinsic whose results are only read
            // an iteration later) down in front of phase B's -- an empty `s_setprio 1; s_setprio 0` was left where the matrix pipe
            // should have been busy, and both waves of a SIMD ran their products in the same interval: an e4m3 K tile took 3900
            // cycles against the 16-bit tile's 2400 (profiles/r6_gemm.md 1)
#pragma unroll
            for (int mt = 0; mt < 4; mt++)
                asm volatile("" : "+v"(acc[mq * 4 + mt][0]), "+v"(acc[mq * 4 + mt][1]), "+v"(acc[mq * 4 + mt][2]), "+v"(acc[mq * 4 + mt][3]));
            __builtin_amdgcn_s_setprio(0);
        }
    };
    auto bar_l = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bar();
    };

    unsigned long long *tl = nullptr;
    auto tstamp = [&](int i) {
        if (TL == 1) {
            unsigned long long now;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            if (tl) tl[i] = now;
        }
    };
    auto tl_open = [&](int id) {
        if (TL == 1 && threadIdx.x == 0) {
            tl = g.diag + (long)id * 8;
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            tl[0] = hw;
            tl[6] = xcc;
        }
    };
    GemmArgs ge = g;

    // EC_EPI_*_LN: the tile's 256 row-statistics pairs (2 KiB) travel into an LDS side area behind the staging
    // buffers by DMA, two slots used alternately (the epilogue of tile T reads slot T & 1 while the pairs of tile
    // T + 1 land in the other).  Waves 0 and 1 issue one piece each, BEFORE the tile's first staging piece: an
    // older request can only retire earlier, so every counted wait below keeps its meaning.  Rows past M read a
    // pair of zeros (the descriptor's range check; never used).
    constexpr bool LNS = epi_is_ln(EPI);
    const bool lds_stats = LNS && g.rowstat_stride == 1;
    float *side = reinterpret_cast<float *>(smem + 2 * KT);
    int slot = 0;
    auto issue_stats = [&](int sl) {
        if constexpr (LNS) {
            if (lds_stats && wave < 2) {
                // buffer-addressed like the staging DMA (a FLAT-encoded global_load_lds in front of them made hipcc's
                // wait-count pass answer every wait behind it with vmcnt(0): one in front of each tile's first
                // accumulator write, one in front of the epilogue's first use of bias / column sums -- both also waits
                // for the staging DMA in flight).  The range is the pairs that exist: a pair past row M reads zeros
                // (never used), so the caller's array needs no readable slack.
                const __amdgpu_buffer_rsrc_t rs = tile_rsrc(g.rowstat, (long)g.M * 8);
                int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    rs, (__attribute__((address_space(3))) void *)(reinterpret_cast<unsigned char *>(side) + sl * 2048 + wave * 1024),
                    16, (m0 + wave * 128 + 2 * ln) * 8, 0, 0, 0);
            }
        }
    };

    if constexpr (TL == 2) clock_stamp(g.diag, 0);
    bool carried = false;     // this tile's K tile 0 was waited for at the hand-over from the tile before
    int id = blockIdx.x;
    tl_open(id);
    tstamp(1);
    setup(id);
    issue_stats(0);
    issue(0, 0);
    issue(2, 0);
    issue(3, 0);
    issue(1, 0);
    if (nk > 1) {
        issue(0, 1);
        issue(2, 1);
        issue(3, 1);
        EC_VMCNT(8);
    } else {
        EC_VMCNT(2);
    }
    bar();
    for (;;) {
        tstamp(2);
        if (wm == 1) bar();   // stagger: the second wave row runs one interval behind
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // one K tile: two phases (32 16-bit MFMAs, or 16 e4m3 MFMAs of twice the cycles, each).  SEG == 2 runs the e4m3 tiles
        // in a loop of their own in front of the 16-bit tiles: one MFMA kind per loop body, so that each loop is allocated
        // like the plain kernel's (both kinds behind a branch in ONE body spilled fragments inside the loop)
        auto ktile = [&](int t, auto f8tag) {
            constexpr bool F8 = decltype(f8tag)::value;
            const int buf = t & 1, nxt = buf ^ 1;
            const bool has1 = t + 1 < nk, has2 = t + 2 < nk;
            const int sc8 = F8 ? (t < (nk8 >> (g.nf8 >> 1)) ? g.f8_scale[0] : g.f8_scale[1]) : 0;
            // ---- phase A ----
            if constexpr (F8) {
                load_m8(buf, 0);
                load_n8(gn0, buf, 0);
                load_n8(gn1, buf, 1);
            } else {
                load_m(buf, 0);
                load_n(fn0, buf, 0);
                load_n(fn1, buf, 1);
            }
            if (has1) {
                issue(1, nxt);
                // behind a tile hand-over every piece of K tile 0 has landed already (waited for before the hand-over
                // barrier), and what is in flight besides K tile 1 are the last epilogue stores: no wait here, the
                // counted wait of phase B (which K tile 1 needs anyway) is the first one they are older than
                if (t > 0 || !carried) EC_VMCNT(8);
            } else {
                EC_VMCNT(0);
                // (satisfied already; tells hipcc's wait-count pass that no staging DMA is pending behind the main
                // loop, so that it does not put a vmcnt(0) of its own in front of the epilogue's first LDS read)
                __builtin_amdgcn_s_waitcnt(0x0F70);
                if constexpr (EPI == EC_EPI_RESID_HL && (HLM & 256) != 0) {
                    // A / B (diagnostic variant 38, round 6): touch the wave tile's residual planes one K tile ahead of the
                    // epilogue that reads them -- one dword of each 128-byte line by LDS-DMA into a dump area (no register
                    // destination; issued as inline asm: requests OLDER than anything the epilogue issues only make its
                    // counted waits conservative), so that the epilogue's loads find the lines in L2 / on their way
                    int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
                    const int tm0 = m0 + wm * 128, tn0 = n0 + wn * 64;
                    const long torg = ((long)tm0 * g.ldc + tn0) * 2, tspan = tile_span(g.M, tm0, 128, g.ldc * 2);
                    const __amdgpu_buffer_rsrc_t th = tile_rsrc(reinterpret_cast<const char *>(g.C) + torg, tspan);
                    const __amdgpu_buffer_rsrc_t tlo = tile_rsrc(reinterpret_cast<const char *>(g.aux) + torg, tspan);
                    const int dump = 2 * KT + 4608 + wave * 256;
                    const int v0 = tn0 < g.N ? ln * (int)g.ldc * 2 : BUF_OOB, v1 = tn0 < g.N ? (ln + 64) * (int)g.ldc * 2 : BUF_OOB;
                    asm volatile("s_mov_b32 m0, %4\n\t"
                                 "buffer_load_dword %0, %2, 0 offen lds\n\t"
                                 "buffer_load_dword %1, %2, 0 offen lds\n\t"
                                 "buffer_load_dword %0, %3, 0 offen lds\n\t"
                                 "buffer_load_dword %1, %3, 0 offen lds"
                                 :: "v"(v0), "v"(v1), "s"(th), "s"(tlo), "s"(dump) : "memory");
                }
            }
            bar_l();
            if constexpr (F8) mma2_g(0, 0, gn0, 1, gn1, sc8);
            else mma2(0, 0, fn0, 1, fn1);
            bar();
            // ---- phase B ----
            if constexpr (F8) load_m8(buf, 1);
            else load_m(buf, 1);
            if (has2) {
                issue(0, buf);
                issue(2, buf);
                issue(3, buf);
                EC_VMCNT(8);
            } else if (has1) {
                EC_VMCNT(2);
            }
            bar_l();
            if constexpr (F8) mma2_g(1, 1, gn1, 0, gn0, sc8);
            else mma2(1, 1, fn1, 0, fn0);
            bar();
        };
        int t = 0;
        if constexpr (SEG == 2)
            for (; t < nk8; t++) ktile(t, std::true_type{});
        for (; t < nk; t++) ktile(t, std::false_type{});
        if (wm == 0) bar();   // balance the stagger barrier: every wave is out of the staging buffers

        const int cm0 = m0, cn0 = n0;
        if (g.splits > 1)
            ge.C = static_cast<unsigned char *>(g.C) + (long)sp0 * g.split_stride * (epi_is16(EPI) ? 2 : 4);
        const int next = id + gridDim.x;
        const bool more = next < ntiles;
        tstamp(3);
        // the next tile's first K tile is requested from INSIDE the epilogue, behind its first loads (epilogue_*_buf)
        auto next_tile = [&]() {
            if (more) {
                setup(next);
                issue_stats(slot ^ 1);
                issue(0, 0);
                issue(2, 0);
                issue(3, 0);
                issue(1, 0);
            }
        };
        const int wm0 = cm0 + wm * 128, wn0 = cn0 + wn * 64;
        // the lane id as the epilogues see it is re-derived per tile and opaque to the optimiser: otherwise every
        // per-lane constant of the epilogue's addressing is hoisted above the tile loop, does not fit beside the main
        // loop's 249 registers and comes back from scratch -- behind an s_waitcnt vmcnt(0) that also waits for the
        // next tile's first K tile, issued a moment ago
        int elane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(elane));
        // Hand-over wait.  vmcnt retires in issue order (loads, stores and LDS-DMA alike) and the DMA of the next
        // tile's K tile 0 is OLDER than everything the epilogue issued, so waiting for all but the epilogue's last TAIL
        // operations is waiting for it -- without waiting for the acknowledgement of the last stores (0.3 - 1.5 k
        // cycles per tile).  TAIL <= the number of operations the buffer-addressed epilogues issue, whatever the
        // tile (their stores are never predicated: the range check drops what does not exist).  The asm form is the
        // scheduling fence; the builtin right behind it costs nothing (it is satisfied already) and tells hipcc's
        // wait-count pass what has completed.
        constexpr bool BUF_EPI = EPI == EC_EPI_RESID_HL || EPI == EC_EPI_STORE16 || EPI == EC_EPI_GELU16 || epi_is_ln(EPI) ||
                                 EPI == EC_EPI_STORE32 || EPI == EC_EPI_RESID32;
        constexpr bool LOUT = (EPI == EC_EPI_STORE16 || EPI == EC_EPI_GELU16) && (HLM & 16) != 0;   // 16-bit output as hi + lo parts (args.aux)
        constexpr bool LOUT8 = LOUT && (HLM & 8) != 0;                                              // ... the lo part as e4m3 bytes
        constexpr int TAIL_HL = ((HLM & 32) && (HLM & 128)) ? 0 : (HLM & (32 | 64 | 128)) ? 16 : 48;     // (diagnostic forms issue fewer operations)
        constexpr int TAIL = !BUF_EPI ? 0 : EPI == EC_EPI_RESID_HL ? TAIL_HL : EPI == EC_EPI_STORE32 ? 32 : EPI == EC_EPI_RESID32 ? 48 : LOUT ? 32 : 16;
        constexpr int enc_tail = (TAIL & 15) | (7 << 4) | (15 << 8) | ((TAIL >> 4) << 14);
        if constexpr (EPI == EC_EPI_RESID_HL)
            epilogue_hl_buf<DT, 8, HLM>(ge, acc, wm0, wn0, elane,
                                        reinterpret_cast<float *>(smem + KT) + wave * ((HLM & 1) ? 2 * 16 * 68 : 16 * 68), next_tile);
        else if constexpr (!epi_is16(EPI))
            epilogue32_buf<EPI, 8, TN>(ge, acc, wm0, wn0, elane, reinterpret_cast<float *>(smem + KT) + wave * (16 * 68), next_tile);
        else if constexpr (BUF_EPI) {
            if (lds_stats)
                epilogue16_buf<DT, EPI, 8, true, (HLM & 4) != 0, LOUT, LOUT8>(ge, acc, wm0, wn0, elane, smem + KT + wave * (2 * 16 * 144),
                                                 side + slot * 512 + wm * 256, next_tile);
            else
                epilogue16_buf<DT, EPI, 8, false, (HLM & 4) != 0, LOUT, LOUT8>(ge, acc, wm0, wn0, elane, smem + KT + wave * (2 * 16 * 144), nullptr, next_tile);
        }
        else {    // the training epilogues (second output / second input): the general form
            next_tile();
            epilogue16_lds<DT, EPI, 8>(ge, acc, wm0, wn0, elane, smem + KT + wave * (2 * 16 * 144), nullptr);
        }
        tstamp(4);
        if (more) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TAIL) : "memory");
            __builtin_amdgcn_s_waitcnt(enc_tail);
        }
        slot ^= 1;
        if (!more) break;
        carried = nk > 1;
        tstamp(5);
        bar();
        if (nk > 1) {
            issue(0, 1);
            issue(2, 1);
            issue(3, 1);
        }
        id = next;
        tl_open(id);
        tstamp(1);
    }
    if (TL == 1) {
        EC_VMCNT(0);
        tstamp(5);
    }
    if constexpr (TL == 2) clock_stamp(g.diag, 2);
}

template <int DT, int EPI, int TL = 0, bool TN = false, int HLM = HL_MODE_DEFAULT, int SEG = 0>
int launch2pp(const GemmArgs &g0, hipStream_t stream)
{
    GemmArgs g = g0;
    g.tiles_m = ec::ceil_div(g.M, 256);
    g.tiles_n = ec::ceil_div(g.N, 256);
    // two staging buffers + the row-statistics side area (LN epilogues) / the tail of the hi-lo epilogue's double
    // scratch (8 waves x 2 x 16 rows x 68 floats = 68 KiB from the second staging buffer on)
    constexpr int lds = 2 * 4 * 128 * 128 + (epi_is_ln(EPI) ? 2 * 2048 : 0) + (EPI == EC_EPI_RESID_HL ? 4608 : 0) +
                        (EPI == EC_EPI_RESID_HL && (HLM & 256) ? 2048 : 0);      // (+ the dump area of the prefetch A / B)
    auto kern = gemm2pp_kernel<DT, EPI, TL, TN, HLM, SEG>;
    if (int rc = ec::ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds)) return rc;
    const int cus = ec::cu_count();
    EC_REQUIRE(cus > 0, "ec_gemm: cannot read the device's compute-unit count");
    constexpr int cls = (EPI == EC_EPI_STORE16 || EPI == EC_EPI_GELU_BWD16 || EPI == EC_EPI_STORE16_LN) ? ec::PROF_GEMM_STORE16
                        : (EPI == EC_EPI_GELU16 || EPI == EC_EPI_GELU16_SAVE || EPI == EC_EPI_GELU16_LN) ? ec::PROF_GEMM_GELU16
                        : (EPI == EC_EPI_RESID32 || EPI == EC_EPI_RESID_HL) ? ec::PROF_GEMM_RESID32 : ec::PROF_GEMM_STORE32;
    constexpr double out_b = (EPI == EC_EPI_STORE16 || EPI == EC_EPI_GELU16 || epi_is_ln(EPI)) ? (!epi_is_ln(EPI) && (HLM & 16) ? 4.0 : 2.0)
                             : (EPI == EC_EPI_GELU16_SAVE || EPI == EC_EPI_GELU_BWD16) ? 4.0
                             : ((EPI == EC_EPI_RESID32 || EPI == EC_EPI_RESID_HL) ? 8.0 : 4.0);
    // segments: every product's flops; the bytes of the parts that exist (a part shared by two segments counts once)
    const int nseg = SEG ? g.nseg : 1;
    const unsigned full = (1u << nseg) - 1, ma = g.seg_mask & full, mw = (g.seg_mask >> 4) & full;
    const int parts_a = SEG && ma != 0 && ma != full ? 2 : 1, parts_w = SEG && mw != 0 && mw != full ? 2 : 1;
    const int nf8 = SEG == 2 ? g.nf8 : 0;    // (an e4m3 product: the same flops, half the operand bytes)
    ec::ProfScope prof(g.splits > 1 ? (int)ec::PROF_GEMM_DW : cls, stream, 2.0 * g.M * g.N * g.K * g.splits * nseg,
                       (2.0 * g.M * g.K * parts_a + 2.0 * g.N * g.K * parts_w + nf8 * (1.0 * g.M * g.K + 1.0 * g.N * g.K) +
                        out_b * g.M * g.N) * g.splits);
    const int tiles = g.tiles_m * g.tiles_n * g.splits;
    hipLaunchKernelGGL(kern, dim3(tiles < cus ? tiles : cus), dim3(512), lds, stream, g);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

#ifdef EC_GEMM_DIAG
#include "gemm_diag.inc"
#endif


// Variants that ship: the default and three independent tilings (kept as cross-checks of each
// other in the GPU tests).  Everything else -- timing experiments that compute wrong results on
// purpose, stamp / timeline builds that write s_memtime records through args.diag, retired A/B
// schedules -- only exists in a -DEC_GEMM_DIAG build (python -m eventclip_amd.build --diag), which
// tools/ load explicitly; the product library rejects those variant numbers.
template <int DT, int EPI> int dispatch_variant(const GemmArgs &g, int variant, hipStream_t s)
{
    switch (variant) {
    case 0: return launch2pp<DT, EPI>(g, s);                     // default: persistent staggered 2-phase
#ifdef EC_GEMM_DIAG
    // the kernels the default one grew out of, kept for comparisons in the diagnostic build only: nothing in the
    // product calls them, and the plain 128 x 128 one once returned five wrong elements in ~10^9 on one box
    case 1: return launch<DT, 256, 256, 2, 4, EPI>(g, s);        // plain two-barrier loop, 256 x 256
    case 2: return launch<DT, 128, 128, 2, 2, EPI>(g, s);
    case 3: return launch<DT, 128, 256, 1, 4, EPI>(g, s);
    case 5: return launch2p<DT, EPI>(g, s);                      // staggered 2-phase, one tile per workgroup
    case 4: return launch4p<DT, EPI>(g, s);
    case 6: return launch2p<DT, EPI, 1>(g, s);   // timing experiment: no DMA in the loop (wrong results)
    case 7: return launch2p<DT, EPI, 2>(g, s);   // timing experiment: every WG streams tile (0,0)
    case 8: return launch2p<DT, EPI, 3>(g, s);   // no s_setprio
    case 9: return launch2p<DT, EPI, 4>(g, s);   // priority on the load segments
    case 10: return launch2p<DT, EPI, 5>(g, s);  // s_memtime stamps of one workgroup -> args.diag
    case 11: return launch2p<DT, EPI, 6>(g, s);
    case 12: return launch_b2<DT, EPI>(g, s);     // two 4-wave workgroups per CU, 128x256x32
    case 13: return launch_b2p<DT, EPI>(g, s);    // same with register-prefetched fragments
    case 14: return launch2p<DT, EPI, 7>(g, s);   // first-round start times spread over a tile period
    case 15: return launch2p<DT, EPI, 8>(g, s);   // ... over half a period
    case 16: return launch2p<DT, EPI, 9>(g, s);   // per-workgroup timeline -> args.diag
    case 18: return launch2pp<DT, EPI, true>(g, s);  // persistent, with timeline records -> args.diag
    case 19: return launch2pp<DT, EPI, 2>(g, s);     // persistent, clock stamps at the workgroup's start and end -> args.diag
    case 43: return launch2pp<DT, EPI, false, false, 5>(g, s);   // 16-bit epilogues: next tile's DMA requested in FRONT of the first use of bias (A / B)
    case 20:                                         // probe: four waves x 128 x 128, 16-bit store only
        if constexpr (EPI == EC_EPI_STORE16) return launch4w<DT>(g, s);
        return ec::fail(EC_ERR_INVALID, "ec_gemm variant 20: store16 only");
    case 21:                                         // ... ring of four quarter tiles, reads and DMA spread through the MFMAs
        if constexpr (EPI == EC_EPI_STORE16) return launch4i<DT>(g, s);
        return ec::fail(EC_ERR_INVALID, "ec_gemm variant 21: store16 only");
#endif
    default:
        return ec::fail(EC_ERR_INVALID, "ec_gemm: unknown variant %d (diagnostic variants need an "
                        "EC_GEMM_DIAG build)", variant);
    }
}

template <int DT> int dispatch_epi(const GemmArgs &g, int epi, int variant, hipStream_t s)
{
    constexpr int HLO = HL_MODE_DEFAULT | 16;    // 16-bit epilogues that also write the lo part (args.aux)
    if (g.nseg > 1) {
        // split-precision operands: the segmented main loop (default variant only)
        EC_REQUIRE(variant == 0, "ec_gemm: A_lo / W_lo need variant 0");
        if (g.nf8 > 0) {
            // ... with e4m3 lo products in front (f16 only; the epilogues the tolerance mode's blocks use)
            if constexpr (DT == EC_F16) {
                switch (epi) {
                case EC_EPI_STORE16:
                    return g.aux ? launch2pp<DT, EC_EPI_STORE16, false, false, HLO, 2>(g, s)
                                 : launch2pp<DT, EC_EPI_STORE16, false, false, HL_MODE_DEFAULT, 2>(g, s);
                case EC_EPI_GELU16:
                    if (g.aux && g.aux8_scale != 0.f) return launch2pp<DT, EC_EPI_GELU16, false, false, HLO | 8, 2>(g, s);
                    return g.aux ? launch2pp<DT, EC_EPI_GELU16, false, false, HLO, 2>(g, s)
                                 : launch2pp<DT, EC_EPI_GELU16, false, false, HL_MODE_DEFAULT, 2>(g, s);
                case EC_EPI_STORE32: return launch2pp<DT, EC_EPI_STORE32, false, false, HL_MODE_DEFAULT, 2>(g, s);
                case EC_EPI_RESID_HL:
                    EC_REQUIRE(g.aux, "ec_gemm: EC_EPI_RESID_HL needs args.aux (the lo plane)");
                    return launch2pp<DT, EC_EPI_RESID_HL, false, false, HL_MODE_DEFAULT, 2>(g, s);
                default: return ec::fail(EC_ERR_INVALID, "ec_gemm: e4m3 lo products go with the STORE16, GELU16, STORE32 and RESID_HL epilogues (got %d)", epi);
                }
            } else {
                return ec::fail(EC_ERR_INVALID, "ec_gemm: e4m3 lo products need dtype EC_F16");
            }
        }
        switch (epi) {
        case EC_EPI_STORE16:
            return g.aux ? launch2pp<DT, EC_EPI_STORE16, false, false, HLO, 1>(g, s)
                         : launch2pp<DT, EC_EPI_STORE16, false, false, HL_MODE_DEFAULT, 1>(g, s);
        case EC_EPI_GELU16:
            return g.aux ? launch2pp<DT, EC_EPI_GELU16, false, false, HLO, 1>(g, s)
                         : launch2pp<DT, EC_EPI_GELU16, false, false, HL_MODE_DEFAULT, 1>(g, s);
        case EC_EPI_STORE32: return launch2pp<DT, EC_EPI_STORE32, false, false, HL_MODE_DEFAULT, 1>(g, s);
        case EC_EPI_RESID32: return launch2pp<DT, EC_EPI_RESID32, false, false, HL_MODE_DEFAULT, 1>(g, s);
        case EC_EPI_RESID_HL:
            EC_REQUIRE(g.aux, "ec_gemm: EC_EPI_RESID_HL needs args.aux (the lo plane)");
            return launch2pp<DT, EC_EPI_RESID_HL, false, false, HL_MODE_DEFAULT, 1>(g, s);
        default: return ec::fail(EC_ERR_INVALID, "ec_gemm: A_lo / W_lo go with the STORE16, GELU16, STORE32, RESID32 and RESID_HL epilogues (got %d)", epi);
        }
    }
    EC_REQUIRE(!((epi == EC_EPI_STORE16 || epi == EC_EPI_GELU16) && g.aux),
               "ec_gemm: STORE16 / GELU16 write their lo part (args.aux) in the launches that take A_lo / W_lo only");
    switch (epi) {
    case EC_EPI_STORE16: return dispatch_variant<DT, EC_EPI_STORE16>(g, variant, s);
    case EC_EPI_GELU16: return dispatch_variant<DT, EC_EPI_GELU16>(g, variant, s);
    case EC_EPI_RESID32: return dispatch_variant<DT, EC_EPI_RESID32>(g, variant, s);
    case EC_EPI_STORE32: return dispatch_variant<DT, EC_EPI_STORE32>(g, variant, s);
    // the training epilogues only exist in the default kernel
    case EC_EPI_GELU16_SAVE:
        EC_REQUIRE(variant == 0 && g.aux, "ec_gemm: EC_EPI_GELU16_SAVE needs variant 0 and args.aux");
        return launch2pp<DT, EC_EPI_GELU16_SAVE>(g, s);
    case EC_EPI_GELU_BWD16:
        EC_REQUIRE(variant == 0 && g.aux, "ec_gemm: EC_EPI_GELU_BWD16 needs variant 0 and args.aux");
        return launch2pp<DT, EC_EPI_GELU_BWD16>(g, s);
    // LayerNorm folded into the GEMMs (default kernel only)
    case EC_EPI_RESID_HL:
#ifdef EC_GEMM_DIAG
        if (variant == 18 && g.aux) return launch2pp<DT, EC_EPI_RESID_HL, true>(g, s);   // timeline records -> args.diag
        if (variant == 19 && g.aux) return launch2pp<DT, EC_EPI_RESID_HL, 2>(g, s);      // clock stamps
        // A / B forms of the hi-lo epilogue (epilogue_hl_buf MODE 0 .. 3)
        if (variant == 30 && g.aux) return launch2pp<DT, EC_EPI_RESID_HL, false, false, 0>(g, s);
        if (variant == 31 && g.aux) return launch2pp<DT, EC_EPI_RESID_HL, false, false, 1>(g, s);
        if (variant == 32 && g.aux) return launch2pp<DT, EC_EPI_RESID_HL, false, false, 2>(g, s);
        if (variant == 33 && g.aux) return launch2pp<DT, EC_EPI_RESID_HL, false, false, 3>(g, s);
        // cost-splitting forms (wrong results on purpose; tools/bench_resid_split.py): no residual loads / no lo store /
        // no store at all / neither loads nor stores
        if (variant == 34 && g.aux) return launch2pp<DT, EC_EPI_RESID_HL, false, false, HL_MODE_DEFAULT | 32>(g, s);
        if (variant == 35 && g.aux) return launch2pp<DT, EC_EPI_RESID_HL, false, false, HL_MODE_DEFAULT | 64>(g, s);
        if (variant == 36 && g.aux) return launch2pp<DT, EC_EPI_RESID_HL, false, false, HL_MODE_DEFAULT | 128>(g, s);
        if (variant == 37 && g.aux) return launch2pp<DT, EC_EPI_RESID_HL, false, false, HL_MODE_DEFAULT | 32 | 128>(g, s);
        if (variant == 38 && g.aux) return launch2pp<DT, EC_EPI_RESID_HL, false, false, HL_MODE_DEFAULT | 256>(g, s);   // residual planes touched a K tile ahead
        if (variant == 13 && g.aux) return launch_b2p<DT, EC_EPI_RESID_HL>(g, s);   // two 4-wave workgroups per CU
#endif
        EC_REQUIRE(variant == 0 && g.aux, "ec_gemm: EC_EPI_RESID_HL needs variant 0 and args.aux (the lo plane)");
        return launch2pp<DT, EC_EPI_RESID_HL>(g, s);
    case EC_EPI_STORE16_LN:
#ifdef EC_GEMM_DIAG
        if (variant == 18 && g.rowstat && g.colsum) return launch2pp<DT, EC_EPI_STORE16_LN, true>(g, s);
        if (variant == 19 && g.rowstat && g.colsum) return launch2pp<DT, EC_EPI_STORE16_LN, 2>(g, s);
        if (variant == 43 && g.rowstat && g.colsum) return launch2pp<DT, EC_EPI_STORE16_LN, false, false, 5>(g, s);
#endif
        EC_REQUIRE(variant == 0 && g.rowstat && g.colsum, "ec_gemm: EC_EPI_STORE16_LN needs variant 0, row_stats and col_sums");
        return launch2pp<DT, EC_EPI_STORE16_LN>(g, s);
    case EC_EPI_GELU16_LN:
#ifdef EC_GEMM_DIAG
        if (variant == 18 && g.rowstat && g.colsum) return launch2pp<DT, EC_EPI_GELU16_LN, true>(g, s);
        if (variant == 19 && g.rowstat && g.colsum) return launch2pp<DT, EC_EPI_GELU16_LN, 2>(g, s);
        if (variant == 43 && g.rowstat && g.colsum) return launch2pp<DT, EC_EPI_GELU16_LN, false, false, 5>(g, s);
#endif
        EC_REQUIRE(variant == 0 && g.rowstat && g.colsum, "ec_gemm: EC_EPI_GELU16_LN needs variant 0, row_stats and col_sums");
        return launch2pp<DT, EC_EPI_GELU16_LN>(g, s);
    default: return ec::fail(EC_ERR_INVALID, "ec_gemm: unknown epilogue %d", epi);
    }
}

// ---------------------------------------------------------------------------------------
// Low-latency form of an under-filled launch.  With a handful of tiles on 256 CUs every workgroup streams all
// of K on its own and the launch takes as long as one tile's K loop (about 1 us per K tile whatever the row
// count: one frame of ViT-L/14 spends 4.6 ms in 96 such launches).  When the caller provides scratch, the product
// is cut into K-batches -- tiles x batches workgroups, each a fraction of K, fp32 partial sums in the scratch --
// and this kernel adds the partial sums up and applies the epilogue.  The fp32 summation order then depends on
// the batch count, i.e. on M: callers that need results independent of the batch size do not pass scratch.
// ---------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void kbatch_fixup_kernel(const float *partial, int splits, int M, int N, const float *bias,
                                                           const float *resid, void *C, long ldc, void *aux, int epi)
{
    typedef typename T16<DT>::elem elem;
    typedef typename T16<DT>::v4 v4;
    const long n4 = (long)M * N / 4, plane = (long)M * N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long e = i * 4, m = e / N;
        const int n = (int)(e - m * N);
        f32x4 a = *reinterpret_cast<const f32x4 *>(partial + e);
        for (int p = 1; p < splits; p++) a += *reinterpret_cast<const f32x4 *>(partial + p * plane + e);
        if (bias) a += *reinterpret_cast<const f32x4 *>(bias + n);
        const long o = m * ldc + n;
        if (epi == EC_EPI_STORE32 || epi == EC_EPI_RESID32) {
            if (epi == EC_EPI_RESID32) a += *reinterpret_cast<const f32x4 *>((resid ? resid : (const float *)C) + o);
            *reinterpret_cast<f32x4 *>((float *)C + o) = a;
            continue;
        }
        v4 out;
        if (epi == EC_EPI_GELU16 || epi == EC_EPI_GELU16_SAVE) {
            if (epi == EC_EPI_GELU16_SAVE) {
                const v4 u = {to16(a[0], elem()), to16(a[1], elem()), to16(a[2], elem()), to16(a[3], elem())};
                *reinterpret_cast<v4 *>((elem *)aux + o) = u;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) out[j] = to16(quick_gelu(a[j]), elem());
        } else if (epi == EC_EPI_GELU_BWD16) {
            const v4 u = *reinterpret_cast<const v4 *>((const elem *)aux + o);
#pragma unroll
            for (int j = 0; j < 4; j++) out[j] = to16((float)to16(a[j], elem()) * quick_gelu_grad((float)u[j]), elem());
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) out[j] = to16(a[j], elem());
        }
        *reinterpret_cast<v4 *>((elem *)C + o) = out;
    }
}

// -> EC_OK after launching the K-batched form, KBATCH_NOT_APPLICABLE (a positive value: never an error code) when the
// launch does not qualify and the caller goes on as usual, or the negative error code of a launch that failed
constexpr int KBATCH_NOT_APPLICABLE = 1;
template <int DT> int try_kbatched(const GemmArgs &g0, int epi, float *ws, size_t ws_bytes, hipStream_t stream)
{
    const int cus = ec::cu_count();
    const int tiles = ec::ceil_div(g0.M, 256) * ec::ceil_div(g0.N, 256);
    if (cus <= 0 || tiles * 2 > cus || g0.N % 4 != 0) return KBATCH_NOT_APPLICABLE;
    const int nk = g0.K / BK;
    int splits = 1;
    while (splits < 16 && tiles * splits * 2 <= cus && nk % (splits * 2) == 0 && nk / (splits * 2) >= 2) splits *= 2;
    if (splits < 2 || (size_t)splits * g0.M * g0.N * 4 > ws_bytes) return KBATCH_NOT_APPLICABLE;
    GemmArgs g = g0;
    g.K = g0.K / splits, g.splits = splits, g.split_stride = (long)g0.M * g0.N;
    g.C = ws, g.ldc = g0.N, g.bias = nullptr, g.resid = nullptr, g.aux = nullptr;
    if (int rc = launch2pp<DT, EC_EPI_STORE32>(g, stream)) return rc;
    const long n4 = (long)g0.M * g0.N / 4;
    hipLaunchKernelGGL(kbatch_fixup_kernel<DT>, dim3((unsigned)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048)), dim3(256),
                       0, stream, ws, splits, g0.M, g0.N, g0.bias, g0.resid, g0.C, g0.ldc, g0.aux, epi);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

}  // namespace

extern "C" EC_API int ec_gemm(const ec_gemm_args *a, ec_stream_t stream)
{
    EC_REQUIRE(a != nullptr, "ec_gemm: args is null");
    EC_REQUIRE(a->M >= 0 && a->N > 0 && a->K > 0, "ec_gemm: bad shape %d x %d x %d", a->M, a->N, a->K);
    if (a->M == 0) return EC_OK;
    EC_REQUIRE(a->K % BK == 0, "ec_gemm: K=%d must be a multiple of %d", a->K, BK);
    EC_REQUIRE(a->N % 16 == 0, "ec_gemm: N=%d must be a multiple of 16", a->N);
    EC_REQUIRE(a->A && a->W && a->C, "ec_gemm: null buffer");
    const long lda = a->lda ? a->lda : a->K, ldc = a->ldc ? a->ldc : a->N;
    EC_REQUIRE(lda % 8 == 0 && ldc % 8 == 0, "ec_gemm: lda/ldc must be multiples of 8 elements");
    // a tile's rows are addressed with 32-bit byte offsets from the tile's origin (256 rows x stride x 4 bytes)
    EC_REQUIRE(lda < (1L << 21) && ldc < (1L << 21) && a->ldw < (1L << 21), "ec_gemm: row strides must be below 2^21 elements");
    EC_REQUIRE((((uintptr_t)a->A | (uintptr_t)a->W | (uintptr_t)a->C) & 15) == 0,
               "ec_gemm: buffers must be 16-byte aligned");
    GemmArgs g;
    g.M = a->M, g.N = a->N, g.K = a->K;
    g.A = a->A, g.lda = lda, g.W = a->W, g.bias = a->bias, g.C = a->C, g.ldc = ldc;
    g.tiles_m = g.tiles_n = 0;
    g.diag = static_cast<unsigned long long *>(a->diag);
    g.ldw = a->ldw ? a->ldw : a->K, g.resid = a->resid, g.aux = a->aux;
    g.splits = a->splits > 1 ? a->splits : 1, g.split_stride = a->split_stride;
    g.tn = a->transposed ? 1 : 0, g.k_valid = a->k_rows;
    g.rowstat = a->row_stats, g.rowstat_stride = a->row_stats_stride > 0 ? a->row_stats_stride : 1, g.colsum = a->col_sums;
    g.stat_out = nullptr, g.stat_groups = 0;
    g.nseg = 1;
    g.seg_da = g.seg_dw = g.seg_mask = 0;
    g.nf8 = 0;
    g.f8_oa[0] = g.f8_oa[1] = g.f8_ow[0] = g.f8_ow[1] = 0, g.f8_scale[0] = g.f8_scale[1] = 0x7f7f7f7f;
    g.aux8_scale = 0.f;
    if (a->aux_e4m3) {
        EC_REQUIRE(a->epilogue == EC_EPI_GELU16 && a->aux && a->A_lo8 && a->dtype == EC_F16 && a->aux_exp >= -60 && a->aux_exp <= 60,
                   "ec_gemm: aux_e4m3 goes with EC_EPI_GELU16, an aux buffer, e4m3 lo products (A_lo8) and -60 <= aux_exp <= 60");
        g.aux8_scale = ldexpf(1.f, a->aux_exp);
    }
    const bool l8 = a->A_lo8 != nullptr, w8 = a->W_lo8 != nullptr;
    if (a->A_lo || a->W_lo || l8 || w8) {
        // split-precision operands: up to three products into the same accumulators, the small ones first
        // (a_lo . w + a . w_lo + a . w; a_lo . w_lo, ~2^-22 of the result, is left out)
        EC_REQUIRE(!a->transposed && a->splits <= 1 && !a->ws && !a->resid && a->variant == 0,
                   "ec_gemm: A_lo / W_lo take no transposed operands, splits, ws or resid, and variant 0");
        EC_REQUIRE((((uintptr_t)a->A_lo | (uintptr_t)a->W_lo) & 15) == 0, "ec_gemm: A_lo / W_lo must be 16-byte aligned");
        g.seg_da = a->A_lo ? (long)((intptr_t)a->A_lo - (intptr_t)a->A) : 0;
        g.seg_dw = a->W_lo ? (long)((intptr_t)a->W_lo - (intptr_t)a->W) : 0;
        int n = 0;
        if (l8 || w8) {
            // e4m3 lo products (the FIRST segments): A_lo8 . W8 in place of a_lo . w, A8 . W_lo8 in place of a . w_lo
            EC_REQUIRE(a->dtype == EC_F16 && a->K % (2 * BK) == 0, "ec_gemm: e4m3 lo products need dtype EC_F16 and K %% 128 == 0 (K = %d)", a->K);
            EC_REQUIRE(!(l8 && a->A_lo) && !(w8 && a->W_lo), "ec_gemm: a lo product is given as 16-bit (A_lo / W_lo) OR as e4m3 (A_lo8 / W_lo8), not both");
            EC_REQUIRE((!l8 || a->W8) && (!w8 || a->A8), "ec_gemm: A_lo8 needs W8 (the e4m3 copy of W), W_lo8 needs A8 (the e4m3 copy of A)");
            EC_REQUIRE((((uintptr_t)a->A_lo8 | (uintptr_t)a->W8 | (uintptr_t)a->A8 | (uintptr_t)a->W_lo8) & 15) == 0,
                       "ec_gemm: e4m3 operands must be 16-byte aligned");
            auto put = [&](const void *pa, const void *pw, int ea, int ew) {
                const int byte = 127 - ea - ew;          // 2^(byte - 127) on the W side undoes both operands' scales
                if (byte < 1 || byte > 254) return false;
                g.f8_oa[n] = (long)((intptr_t)pa - (intptr_t)a->A), g.f8_ow[n] = (long)((intptr_t)pw - (intptr_t)a->W);
                g.f8_scale[n] = byte * 0x01010101;
                n++;
                return true;
            };
            EC_REQUIRE(!l8 || put(a->A_lo8, a->W8, a->a_lo8_exp, a->w8_exp), "ec_gemm: a_lo8_exp + w8_exp = %d outside -127 .. 126", a->a_lo8_exp + a->w8_exp);
            EC_REQUIRE(!w8 || put(a->A8, a->W_lo8, a->a8_exp, a->w_lo8_exp), "ec_gemm: a8_exp + w_lo8_exp = %d outside -127 .. 126", a->a8_exp + a->w_lo8_exp);
            g.nf8 = n;
        }
        if (a->A_lo) g.seg_mask |= 1u << n, n++;             // a_lo . w
        if (a->W_lo) g.seg_mask |= 1u << (4 + n), n++;       // a . w_lo
        n++;                                                 // a . w
        g.nseg = n;
    }
    // what the epilogues read these with: row_stats by 16-byte LDS-DMA (two pairs at a time at stride 1, through a
    // descriptor whose range ends at pair M - 1: nothing past the array is read), col_sums as float4, row_sums written as float2
    EC_REQUIRE((((uintptr_t)a->row_stats | (uintptr_t)a->col_sums) & 15) == 0, "ec_gemm: row_stats / col_sums must be 16-byte aligned");
    EC_REQUIRE(!a->row_stats || a->M < (1 << 27), "ec_gemm: row_stats addresses its pairs with 32-bit byte offsets (M = %d)", a->M);
    EC_REQUIRE(((uintptr_t)a->row_sums & 7) == 0, "ec_gemm: row_sums must be 8-byte aligned");
    if (a->epilogue == EC_EPI_RESID_HL && a->row_sums) {
        EC_REQUIRE(a->N % 64 == 0, "ec_gemm: row_sums needs N %% 64 == 0 (N = %d)", a->N);
        g.stat_out = a->row_sums, g.stat_groups = a->N / 64;
    }
    if (g.tn) {
        g.lda = a->lda ? a->lda : a->M, g.ldw = a->ldw ? a->ldw : a->N;
        EC_REQUIRE(a->epilogue == EC_EPI_STORE32 && a->variant == 0 && !a->bias && !a->ws,
                   "ec_gemm: transposed operands go with EC_EPI_STORE32, variant 0, no bias, no ws");
        EC_REQUIRE(g.M % 8 == 0 && g.M >= 8 && g.lda % 8 == 0 && g.ldw % 8 == 0 && g.lda >= g.M && g.ldw >= g.N,
                   "ec_gemm: transposed operands need M %% 8 == 0 and row strides (multiples of 8) covering M and N");
        EC_REQUIRE(g.k_valid > 0 && (long)g.k_valid <= (long)g.K * g.splits,
                   "ec_gemm: k_rows=%d outside (0, splits * K = %ld]", g.k_valid, (long)g.K * g.splits);
        EC_REQUIRE(g.splits == 1 || (g.split_stride >= (long)(g.M - 1) * ldc + g.N && g.split_stride % 8 == 0),
                   "ec_gemm: split_stride %ld too small for an %d x %d output", g.split_stride, g.M, g.N);
        hipStream_t st = static_cast<hipStream_t>(stream);
        if (a->dtype == EC_F16) return launch2pp<EC_F16, EC_EPI_STORE32, false, true>(g, st);
        if (a->dtype == EC_BF16) return launch2pp<EC_BF16, EC_EPI_STORE32, false, true>(g, st);
        return ec::fail(EC_ERR_INVALID, "ec_gemm: unknown dtype %d", a->dtype);
    }
    EC_REQUIRE(g.ldw % 8 == 0 && g.ldw >= (long)g.K * g.splits && lda >= (long)g.K * g.splits,
               "ec_gemm: lda / ldw must cover splits * K columns and be multiples of 8 elements");
    EC_REQUIRE((((uintptr_t)a->resid | (uintptr_t)a->aux) & 15) == 0, "ec_gemm: resid / aux must be 16-byte aligned");
    EC_REQUIRE(!a->resid || a->epilogue == EC_EPI_RESID32, "ec_gemm: args.resid goes with EC_EPI_RESID32");
    if (g.ldw != g.K || g.splits > 1 || g.resid) {
        EC_REQUIRE(a->variant == 0, "ec_gemm: ldw / splits / resid need variant 0");
        EC_REQUIRE(g.splits == 1 || (g.split_stride >= (long)(g.M - 1) * ldc + g.N && g.split_stride % 8 == 0),
                   "ec_gemm: split_stride %ld too small for an %d x %d output", g.split_stride, g.M, g.N);
    }
#ifdef EC_GEMM_DIAG
    {
        const int v = a->variant;
        EC_REQUIRE(!(v == 10 || v == 16 || v == 18 || v == 19) || a->diag, "ec_gemm: variant %d needs args.diag", v);
    }
#endif
    hipStream_t s = static_cast<hipStream_t>(stream);
    EC_REQUIRE(a->epilogue < EC_EPI_RESID_HL || (g.splits == 1 && !a->ws && !a->resid),
               "ec_gemm: the folded-LayerNorm epilogues take no splits / ws / resid");
    if (a->ws && a->variant == 0 && g.splits == 1 && a->epilogue >= EC_EPI_STORE16 && a->epilogue <= EC_EPI_GELU_BWD16) {
        EC_REQUIRE(((uintptr_t)a->ws & 15) == 0, "ec_gemm: ws must be 16-byte aligned");
        // what dispatch_epi would check: the fixup kernel reads / writes through these on the device
        EC_REQUIRE(!(a->epilogue == EC_EPI_GELU16_SAVE || a->epilogue == EC_EPI_GELU_BWD16) || a->aux != nullptr,
                   "ec_gemm: epilogue %d needs aux", a->epilogue);
        EC_REQUIRE(a->resid == nullptr || ((uintptr_t)a->resid & 15) == 0, "ec_gemm: resid must be 16-byte aligned");
        int rc = KBATCH_NOT_APPLICABLE;
        if (a->dtype == EC_F16) rc = try_kbatched<EC_F16>(g, a->epilogue, static_cast<float *>(a->ws), a->ws_bytes, s);
        else if (a->dtype == EC_BF16) rc = try_kbatched<EC_BF16>(g, a->epilogue, static_cast<float *>(a->ws), a->ws_bytes, s);
        if (rc != KBATCH_NOT_APPLICABLE) return rc;   // launched, or failed with a real error code
    }
    if (a->dtype == EC_F16) return dispatch_epi<EC_F16>(g, a->epilogue, a->variant, s);
    if (a->dtype == EC_BF16) return dispatch_epi<EC_BF16>(g, a->epilogue, a->variant, s);
    return ec::fail(EC_ERR_INVALID, "ec_gemm: unknown dtype %d", a->dtype);
}

#ifdef EC_GEMM_DIAG
extern "C" EC_API int ec_mfma_probe(int mode, int iters, int workgroups, const void *src, float *out, ec_stream_t stream)
{
    EC_REQUIRE(mode >= 0 && mode < 16 && iters > 0 && workgroups > 0 && src && out, "ec_mfma_probe: bad arguments");
    void (*k[16])(const unsigned char *, float *, int) = {mfma_probe_kernel<0>, mfma_probe_kernel<1>, mfma_probe_kernel<2>, mfma_probe_kernel<3>,
                                                          mfma_probe_kernel<4>, mfma_probe_kernel<5>, mfma_probe_kernel<6>, mfma_probe_kernel<7>,
                                                          mfma_probe_kernel<8>, mfma_probe_kernel<9>, mfma_probe_kernel<10>, mfma_probe_kernel<11>,
                                                          mfma_probe_kernel<12>, mfma_probe_kernel<13>, mfma_probe_kernel<14>, mfma_probe_kernel<15>};
    if (int rc = ec::ensure_dynamic_lds(reinterpret_cast<const void *>(k[mode]), 160 * 1024)) return rc;
    hipLaunchKernelGGL(k[mode], dim3(workgroups), dim3(256), 160 * 1024, static_cast<hipStream_t>(stream),
                       static_cast<const unsigned char *>(src), out, iters);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}
#endif
