// Fused multi-head self-attention for the CLIP towers (gfx950), head dim 64.
//
// Replaces nn.MultiheadAttention inside the ResidualAttentionBlocks of
// un-vendored openai/CLIP clip/model.py (reached through
// models/clip_cls.py:84 encode_text / :101 encode_image).  Sequences are short
// and fixed (50..577 vision tokens, 77 text tokens), so one workgroup owns one
// (sequence, head): the head's whole K and V (<= 608 x 64 x 16 bit each) sit in
// LDS and every wave takes 16-query tiles.  Keys are walked in 64-key blocks with
// an online softmax (running max / sum per query, O rescaled per block: 16
// registers), which keeps the kernel near 100 VGPRs for any sequence length.
//
//  * S^T = K . Q^T with v_mfma_f32_16x16x32: K rows from LDS (ds_read_b128,
//    XOR-swizzled 128-B rows), Q straight from HBM as the B operand.  A lane then
//    holds one query column: softmax reductions are in-lane plus two xor-shuffles.
//  * O^T = V^T . P^T: the exponentiated score registers ARE the B operand (the k
//    index inside each 32-key step is permuted consistently on both operands), so
//    P never moves between lanes or through LDS.  V stays row-major in LDS and is
//    read transposed with ds_read_b64_tr_b16; its 8-byte slots are XOR-swizzled
//    by (row >> 1) & 3 so the transposed reads are bank-conflict free.  The V^T
//    row order is permuted so each lane ends with 16 contiguous head-dim outputs
//    (two 16-B stores).
//  * fp32 scores / softmax / accumulation; the 1/sqrt(64) scale is a power of two.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "mfma.h"
#include "tower_ops.h"

#ifdef EC_ATTN_DIAG
// diagnostic build only (python -m eventclip_amd.build --diag, tools/timeline_attn.py): s_memtime at
// workgroup start / K and V staged / done, and the CU the workgroup ran on
__device__ unsigned long long ec_attn_stamps[4 * 65536];
#endif

namespace {

using namespace ec;

#ifdef EC_ATTN_DIAG
__device__ __forceinline__ void attn_stamp(int i)
{
    unsigned long long now;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    if (threadIdx.x == 0 && blockIdx.x < 65536) ec_attn_stamps[blockIdx.x * 4 + i] = now;
}
#else
__device__ __forceinline__ void attn_stamp(int) {}
#endif

// Waves per workgroup: 8 when two workgroups fit a CU's LDS (K + V <= 80 KiB: S <= 320), 16 when the
// sequence's K and V leave room for one workgroup only (S = 577: 148 KiB) -- four waves per SIMD either way.

struct AttnArgs {
    const void *qkv;  // [n_seq * S, 3W] 16-bit: q | k | v, heads are 64-wide column blocks
    void *out;        // [n_seq * S, W] 16-bit
    int S, W, heads, causal;
    int q_rows;       // only the first q_rows query rows of every sequence are computed; out is
                      // [n_seq * q_rows, W] (q_rows = S: the whole sequence)
    float scale_log2e;
    int q_scaled;     // 1: the q columns already hold q * scale_log2e (QM_INPUT); 2: plain q, scores scaled in fp32 (QM_RAW)
    float *lse;       // LSE kernels only: [n_seq, heads, S] fp32, log2 of the softmax denominator in the
                      // scaled-score domain (m * scale_log2e + log2 l), kept for ec_attention_backward
};

// One 64-key (or, for the last odd step, 32-key) block of the online softmax:
// scores -> running max / sum -> P^T as the B operand -> O^T += V^T . P^T.
template <int DT, int KSTEPS, bool MASK>  // KSTEPS = 32-key steps in this block (2 or 1)
__device__ __forceinline__ void attn_block(const unsigned char *ldsK, const unsigned char *ldsV,
                                           int key0, int klimit, const typename T16<DT>::v8 (&qf)[2],
                                           float scale_log2e, float &m_run, float &l_run,
                                           f32x4 (&o)[4], int g, int c16)
{
    typedef typename T16<DT>::elem elem;
    typedef typename T16<DT>::v8 v8;
    typedef typename T16<DT>::v4 v4;
    constexpr int NT = 2 * KSTEPS;
    // ---- scores: acc[kt][r] = <k[key0 + 16 kt + 4 g + r], q[c16]> ----
    f32x4 acc[NT];
#pragma unroll
    for (int kt = 0; kt < NT; kt++) {
        acc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int row = key0 + kt * 16 + c16;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const v8 kf = *reinterpret_cast<const v8 *>(ldsK + row * 128 +
                                                        (((ks * 4 + g) ^ (row & 7)) << 4));
            acc[kt] = mfma16(kf, qf[ks], acc[kt]);
        }
    }
    // ---- (mask,) block max, rescale factor ----
    float mx = m_run;
#pragma unroll
    for (int kt = 0; kt < NT; kt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (MASK) {   // only blocks that reach past klimit (padding, causal diagonal)
                const int key = key0 + kt * 16 + 4 * g + r;
                acc[kt][r] = key < klimit ? acc[kt][r] : -INFINITY;
            }
            mx = fmaxf(mx, acc[kt][r]);
        }
    mx = xor_max(mx);   // the four lane groups hold different keys of the same query
    const float alpha = __builtin_amdgcn_exp2f((m_run - mx) * scale_log2e);
    m_run = mx;
    const float mc = -mx * scale_log2e;
    float psum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; kt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[kt][r], scale_log2e, mc));
            acc[kt][r] = p;
            psum += p;
        }
    l_run = l_run * alpha + psum;   // per-lane partial; summed over the lane groups at the end
#pragma unroll
    for (int dt = 0; dt < 4; dt++) o[dt] *= alpha;
    // ---- O^T += V^T . P^T ----
#pragma unroll
    for (int s = 0; s < KSTEPS; s++) {
        // B operand element j <-> key key0 + 32 s + 16 (j >> 2) + 4 g + (j & 3)
        v8 pf;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            pf[r] = to16(acc[2 * s][r], elem());
            pf[4 + r] = to16(acc[2 * s + 1][r], elem());
        }
#pragma unroll
        for (int dt = 0; dt < 4; dt++) {
            // A operand row i <-> head dim (i >> 2) * 16 + 4 dt + (i & 3), same key order
            v8 vf;
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const int row = key0 + 32 * s + 16 * hh + 4 * g + (c16 >> 2);
                const int u = ((c16 & 3) * 4 + dt) ^ ((row >> 1) & 3);
                const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4 *)(ldsV + row * 128 + u * 8));
                const v4 tv = __builtin_bit_cast(v4, t);
                vf[4 * hh] = tv[0], vf[4 * hh + 1] = tv[1], vf[4 * hh + 2] = tv[2],
                        vf[4 * hh + 3] = tv[3];
            }
            o[dt] = mfma16(vf, pf, o[dt]);
        }
    }
}

// Version 2 of the block (the product path; `attn_block` above is kept for A/B in the diagnostic build).
// The vector ALU is the saturated pipe of this kernel (profiles/r2_attention.md), so the block is
// arranged to leave it only what no other unit can do -- one exp2, half a max3 and half a convert per score:
//  * the query fragment arrives pre-multiplied by log2(e) / sqrt(64), so the MFMA result is already the
//    exponent's argument up to the row offset;
//  * the row offset is the MFMA's C operand: mneg = {-m, -m, -m, -m} for the lane's query (a lane's four
//    accumulator rows belong to ONE query column), so the matrix pipe delivers s - m and no per-score
//    subtract is left;
//  * m is the running maximum as of the last time it MOVED, not of the block.  It starts at 0; a block whose
//    scores stay below m + ATTN_THR (log2 units; and, for a tile's first block only, whose maximum is not
//    below m - ATTN_LO) keeps m, O and the row sum as they are -- P <= 2^ATTN_THR is exact in either 16-bit
//    type, and nothing is rescaled.  Only when some lane sees a larger score (a wave-uniform branch, seldom
//    taken) are m, the scores of this block, O and the row sum moved to the new maximum, all of them
//    exactly once, BEFORE this block's P is formed (cdna_hip_programming.md T13's safe order);
//  * the row sum comes from the matrix pipe too: a fifth "V^T" tile of ones, so l = sum of the ROUNDED
//    P the output was built from.
// Where the factor c = log2(e) / sqrt(64) of the scores comes from (template parameter QM of the block):
//  QM_INPUT  the q columns of qkv already hold c q: the image tower folds c into the q rows of in_proj_weight /
//            in_proj_bias BEFORE they are rounded to 16 bit (ec_vit_weights.q_scaled), so the product is rounded
//            once, like an unscaled q -- no work and no error in the kernel;
//  QM_KERNEL the kernel multiplies its query fragment by c and rounds it to the operand type a second time: f16
//            callers that pass a plain q (ec_attention, the training forward).  +20 % on the kernel's own
//            rounding error on unit-variance data (6.3e-4 against 4.8e-4), +25 % on the logits of a tower with
//            sharp attention (tests/test_configs_gpu.py configs[3]), which is why the tower uses QM_INPUT;
//  QM_RAW    scores stay in raw units and every score is multiplied by c in fp32: bf16 callers with a plain q
//            (8 bits of a twice-rounded q would put 2e-2 on the log-sum-exp the training forward saves).
enum { QM_RAW = 0, QM_KERNEL = 1, QM_INPUT = 2 };
constexpr float ATTN_THR = 10.f;
constexpr float ATTN_LO = 4.f;   // first block: P of the row maximum >= 2^-4, the rest of a 16-bit float's range below it

// O += A . B IN PLACE (tied operand), one asm statement per MFMA.  Through the builtin hipcc lets the
// loop-carried accumulators wander (D != C) and pays for it with a copy of all of O in every block once the
// block has the seldom-taken branch below in it; an asm statement's "+v" operand cannot move (one statement
// with all five accumulators tied brings the copies back: one each).  What hipcc does not pad for an asm
// statement (cdna_hip_programming.md 5.7 item 2) is handled by hand:
//  pad 1 = two wait states in front -- a just-written vector result needs them before an MFMA reads it (a
//          step's first MFMA: the converted P);
//  pad 2 = twelve behind -- a block's last MFMA: hipcc may read O right behind the block (copies at loop
//          exits, spills, the epilogue) and pads nothing for a producer inside an asm string.
// Elsewhere O arrives from the previous step's MFMAs (an accumulate chain on the whole of D needs none) and the
// V^T fragments from ds_reads (the compiler's own s_waitcnt covers asm inputs).  tools/check_attn_isa.py
// checks the compiled kernels for vector writes right in front of the unpadded statements.
template <int pad> __device__ __forceinline__ void mfma16_acc(const f16x8 &a, const f16x8 &b, f32x4 &c)
{
    if (pad == 1)
        asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    else if (pad == 2)
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 11" : "+v"(c) : "v"(a), "v"(b));
    else
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <int pad> __device__ __forceinline__ void mfma16_acc(const bf16x8 &a, const bf16x8 &b, f32x4 &c)
{
    if (pad == 1)
        asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    else if (pad == 2)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 11" : "+v"(c) : "v"(a), "v"(b));
    else
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

// The seldom-taken arm: move the lane's query to a new maximum.  d = how far in the units of the scores,
// d_log2 = in log2 units (the same where the query arrives pre-scaled); everything still
// at the old m moves with it exactly once: the caller's pending scores (by the caller), O and the row sum
// (scaled by 2^-d; by 0 while they are still empty: `down`), and -m.  O and -m IN PLACE like the MFMAs'
// accumulate; O was last written by the previous block's MFMAs, whose last statement waited for them.
__device__ __forceinline__ void attn_move(float d, float d_log2, bool down, f32x4 &mneg, f32x4 (&o)[5])
{
    float alpha = __builtin_amdgcn_exp2f(-d_log2);
    alpha = down ? 0.f : alpha;
#pragma unroll
    for (int dt = 0; dt < 5; dt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float t = o[dt][r];
            asm volatile("s_nop 0\n\tv_mul_f32 %0, %0, %1" : "+v"(t) : "v"(alpha));   // s_nop: alpha is fresh from v_exp / v_cndmask
            o[dt][r] = t;
        }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        float t = mneg[r];
        asm volatile("v_sub_f32 %0, %0, %1" : "+v"(t) : "v"(d));
        mneg[r] = t;
    }
}

// Position of a key row's 16-byte K chunk c / 8-byte V slot u inside its 128-byte LDS row: IMG 0 = the image of the
// 16-query-tile kernel (conflict-free for ITS fragment reads), IMG 1 = the image of attention32_kernel, whose 32-row
// A-operand reads and transposed reads would be two-way conflicted on image 0 (MI355X_MICROARCH.md, LDS lane groups:
// rows r and r + 24 / r + 8 of a ds_read_b128 group share row & 7; the two key pairs of a transposed read share a
// 64-byte half).  The 16-row block functions take IMG so that attention32_kernel's last-row path can read its image.
template <int IMG> __device__ __forceinline__ int kkey(int row) { return IMG ? (row >> 1) & 7 : row & 7; }
template <int IMG> __device__ __forceinline__ int vkey(int row) { return IMG ? ((row >> 1) & 1) << 3 : (row >> 1) & 3; }

// One block of KSTEPS 32-key steps.  `down`: the tile has added nothing yet (m may move down too).
template <int DT, int KSTEPS, bool MASK, int QM, int IMG = 0>
__device__ __forceinline__ void attn_block2(const unsigned char *ldsK, const unsigned char *ldsV,
                                            int key0, int klimit, const typename T16<DT>::v8 (&qf)[2],
                                            const typename T16<DT>::v8 &ones, bool down, float c, f32x4 &mneg,
                                            f32x4 (&o)[5], int g, int c16)
{
    constexpr bool PRE = QM != QM_RAW;   // else: scores in raw units, c = log2(e) / sqrt(64) applied per score
    const float thr = PRE ? ATTN_THR : ATTN_THR / c, lo = PRE ? ATTN_LO : ATTN_LO / c;
    typedef typename T16<DT>::elem elem;
    typedef typename T16<DT>::v8 v8;
    typedef typename T16<DT>::v4 v4;
    constexpr int NT = 2 * KSTEPS;
    // ---- acc[kt][r] = c <k[key0 + 16 kt + 4 g + r], q[c16]> - m[c16] ----
    f32x4 acc[NT];
#pragma unroll
    for (int kt = 0; kt < NT; kt++) {
        const int row = key0 + kt * 16 + c16;
        const v8 k0 = *reinterpret_cast<const v8 *>(ldsK + row * 128 + (((0 + g) ^ kkey<IMG>(row)) << 4));
        const v8 k1 = *reinterpret_cast<const v8 *>(ldsK + row * 128 + (((4 + g) ^ kkey<IMG>(row)) << 4));
        acc[kt] = mfma16(k0, qf[0], mneg);
        acc[kt] = mfma16(k1, qf[1], acc[kt]);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; kt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (MASK) {   // only blocks that reach past klimit (padding, causal diagonal)
                const int key = key0 + kt * 16 + 4 * g + r;
                acc[kt][r] = key < klimit ? acc[kt][r] : -INFINITY;
            }
            mx = fmaxf(mx, acc[kt][r]);
        }
    if (__builtin_amdgcn_ballot_w64(mx > thr || (down && mx < -lo)) != 0) {
        // the four lane groups hold different keys of the same query and all hold parts of its O column:
        // one distance for all of them; up only, unless nothing has been added yet.  Only the queries that
        // need it move (d = 0, scale 1 for the others: bit for bit as if the branch had not been taken), so a
        // query's result never depends on which other queries share its tile (ec_attention_rows computes a
        // prefix of the rows next to whatever the caller left in the others)
        const float mq = xor_max(mx);
        const float d = (mq > thr || (down && mq < -lo)) ? mq : 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; kt++) acc[kt] -= d;
        attn_move(d, PRE ? d : d * c, down, mneg, o);
    }
    // ---- O^T += V^T . P^T, row sum += 1^T . P^T ----
#pragma unroll
    for (int s = 0; s < KSTEPS; s++) {
        // the step's four V^T fragments are requested first: their LDS latency passes under the
        // exponentials.  A operand row i <-> head dim (i >> 2) * 16 + 4 dt + (i & 3), key order as below
        v8 vf[4];
#pragma unroll
        for (int dt = 0; dt < 4; dt++)
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const int row = key0 + 32 * s + 16 * hh + 4 * g + (c16 >> 2);
                const int u = ((c16 & 3) * 4 + dt) ^ vkey<IMG>(row);
                const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4 *)(ldsV + row * 128 + u * 8));
                const v4 tv = __builtin_bit_cast(v4, t);
                vf[dt][4 * hh] = tv[0], vf[dt][4 * hh + 1] = tv[1], vf[dt][4 * hh + 2] = tv[2],
                            vf[dt][4 * hh + 3] = tv[3];
            }
        // B operand element j <-> key key0 + 32 s + 16 (j >> 2) + 4 g + (j & 3)
        v8 pf;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            pf[r] = to16(__builtin_amdgcn_exp2f(PRE ? acc[2 * s][r] : acc[2 * s][r] * c), elem());
            pf[4 + r] = to16(__builtin_amdgcn_exp2f(PRE ? acc[2 * s + 1][r] : acc[2 * s + 1][r] * c), elem());
        }
        mfma16_acc<1>(ones, pf, o[4]);
#pragma unroll
        for (int dt = 0; dt < 3; dt++) mfma16_acc<0>(vf[dt], pf, o[dt]);
        if (s == KSTEPS - 1)
            mfma16_acc<2>(vf[3], pf, o[3]);
        else
            mfma16_acc<0>(vf[3], pf, o[3]);
    }
}

// A single key (the 257th / 577th token of the ViT sequences: S = 32 n + 1) as a rank-one update instead of
// a masked 32-key step: two MFMAs for its 16 scores -- the LDS rows behind the last key are copies of it, so
// every row of the key tile, hence every accumulator register of a lane, is the lane's query against THIS
// key -- one exp2, and 16 multiply-adds of the key's V row (head dims 16 g .. 16 g + 15 for this lane, as the
// epilogue stores them) instead of 8 masked scores, 5 MFMAs and 16 LDS reads.  P is rounded to 16 bit like
// every other P.  Must follow the tile's last block (its last MFMA waited for O to settle).
template <int DT, int QM, int IMG = 0>
__device__ __forceinline__ void attn_odd_key(const unsigned char *ldsK, const unsigned char *ldsV, int key,
                                             const typename T16<DT>::v8 (&qf)[2], bool down, float c, f32x4 &mneg,
                                             f32x4 (&o)[5], int g, int c16)
{
    typedef typename T16<DT>::elem elem;
    typedef typename T16<DT>::v8 v8;
    typedef typename T16<DT>::v4 v4;
    constexpr bool PRE = QM != QM_RAW;
    const float thr = PRE ? ATTN_THR : ATTN_THR / c, lo = PRE ? ATTN_LO : ATTN_LO / c;
    const int row = key + c16;
    const v8 k0 = *reinterpret_cast<const v8 *>(ldsK + row * 128 + (((0 + g) ^ kkey<IMG>(row)) << 4));
    const v8 k1 = *reinterpret_cast<const v8 *>(ldsK + row * 128 + (((4 + g) ^ kkey<IMG>(row)) << 4));
    v4 vv[4];
#pragma unroll
    for (int dt = 0; dt < 4; dt++)   // logical 8-byte slot 4 g + dt of V's row, stored at slot ^ vkey(row)
        vv[dt] = *reinterpret_cast<const v4 *>(ldsV + key * 128 + ((4 * g + dt) ^ vkey<IMG>(key)) * 8);
    f32x4 acc = mfma16(k0, qf[0], mneg);
    acc = mfma16(k1, qf[1], acc);
    float sc = acc[0];
    if (__builtin_amdgcn_ballot_w64(sc > thr || (down && sc < -lo)) != 0) {
        const float d = (sc > thr || (down && sc < -lo)) ? sc : 0.f;   // the same in the query's four lane groups
        sc -= d;
        attn_move(d, PRE ? d : d * c, down, mneg, o);
    }
    const float p = (float)to16(__builtin_amdgcn_exp2f(PRE ? sc : sc * c), elem());
    o[4][0] += p;    // the epilogue reads register 0 of the row-sum tile
#pragma unroll
    for (int dt = 0; dt < 4; dt++)
#pragma unroll
        for (int r = 0; r < 4; r++) o[dt][r] = __builtin_fmaf(p, (float)vv[dt][r], o[dt][r]);
}

// Keys of one tile, not causal: the full 32-key steps [step0, step1) in blocks of two, then (`tail`) whatever
// lies behind the last full step of the sequence: nothing, one key (rank-one update) or a masked step.
template <int DT, int QM, int IMG = 0>
__device__ __forceinline__ void attn_keys(const unsigned char *ldsK, const unsigned char *ldsV, int S, int step0,
                                          int step1, bool tail, const typename T16<DT>::v8 (&qf)[2],
                                          const typename T16<DT>::v8 &ones, float c, f32x4 &mneg, f32x4 (&o)[5],
                                          int g, int c16)
{
    bool down = true;
    int s = step0;
    for (; s + 2 <= step1; s += 2) {
        attn_block2<DT, 2, false, QM, IMG>(ldsK, ldsV, 32 * s, S, qf, ones, down, c, mneg, o, g, c16);
        down = false;
    }
    if (s < step1) {
        attn_block2<DT, 1, false, QM, IMG>(ldsK, ldsV, 32 * s, S, qf, ones, down, c, mneg, o, g, c16);
        down = false;
    }
    if (tail) {
        const int full = S >> 5, nt = S - 32 * full;
        if (nt == 1)
            attn_odd_key<DT, QM, IMG>(ldsK, ldsV, S - 1, qf, down, c, mneg, o, g, c16);
        else if (nt > 1)
            attn_block2<DT, 1, true, QM, IMG>(ldsK, ldsV, 32 * full, S, qf, ones, down, c, mneg, o, g, c16);
    }
}

// Floats per wave in the merge area of a tile whose keys are split over the waves: O (64), m, l
constexpr int ATTN_PART = 66;
// Is the sequence's LAST query tile (a single row, S = 16 n + 1) split over the waves?  ONE predicate for the kernel (which
// writes the merge area behind the K / V images) and for the host (which reserves it): depends on S, the mask and the wave
// count alone, never on q_rows.
__host__ __device__ constexpr bool attn_lone_tile(int S, int causal, int waves)
{
    return !causal && (S & 15) == 1 && (S + 15) / 16 > waves && ((S + 15) / 16) % waves == 1;
}

template <int DT, int AT_WAVES, bool LSE = false, bool V2 = true, int QM = (DT == 0 ? QM_KERNEL : QM_RAW)>
__global__ __launch_bounds__(AT_WAVES * 64, 4) void attention_kernel(const AttnArgs a)
{
    typedef typename T16<DT>::elem elem;
    typedef typename T16<DT>::v8 v8;

    constexpr int AT_THREADS = AT_WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int S = a.S, W = a.W;
    const int n32 = (S + 31) / 32;       // 32-key steps; keys padded to 32 * n32
    const int SP = 32 * n32;
    unsigned char *ldsK = smem;
    unsigned char *ldsV = smem + SP * 128;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g_lane = lane >> 4, c_lane = lane & 15;
    const int head = blockIdx.x % a.heads, seq = blockIdx.x / a.heads;
    const long ld = 3L * W;
    const elem *base = (const elem *)a.qkv + (long)seq * S * ld + head * 64;
    attn_stamp(0);

    // ---- stage K and V of this head: AT_THREADS / 8 rows x 8 chunks of 16 B per pass, at most five passes
    // (S <= 320 on 8 waves, <= 640 on 16).  V2: every load of the head is requested before the first LDS
    // write, so the staging costs one memory round trip instead of one per pass (a third of a workgroup's
    // life at S = 257 otherwise) ----
    if constexpr (V2) {
        const int r_in = threadIdx.x >> 3, ch = threadIdx.x & 7;
        u32x4 kv[5], vv[5];
#pragma unroll
        for (int p = 0; p < 5; p++) {
            const int row = r_in + p * (AT_THREADS / 8);
            if (row < SP) {
                const int srow = row < S ? row : S - 1;  // padded keys: finite data, masked below
                const elem *src = base + (long)srow * ld + ch * 8;
                kv[p] = *reinterpret_cast<const u32x4 *>(src + W);
                vv[p] = *reinterpret_cast<const u32x4 *>(src + 2 * W);
            }
        }
#pragma unroll
        for (int p = 0; p < 5; p++) {
            const int row = r_in + p * (AT_THREADS / 8);
            if (row < SP) {
                *reinterpret_cast<u32x4 *>(ldsK + row * 128 + ((ch ^ (row & 7)) << 4)) = kv[p];
                // V: 8-byte slot u -> u ^ ((row>>1)&3): chunk moves by bit 1, halves swap by bit 0
                u32x4 t = vv[p];
                if ((row >> 1) & 1) t = u32x4{t[2], t[3], t[0], t[1]};
                *reinterpret_cast<u32x4 *>(ldsV + row * 128 + ((ch ^ ((row >> 2) & 1)) << 4)) = t;
            }
        }
    } else {
        const int r_in = threadIdx.x >> 3, ch = threadIdx.x & 7;
        for (int row = r_in; row < SP; row += AT_THREADS / 8) {
            const int srow = row < S ? row : S - 1;  // padded keys: finite data, masked below
            const elem *src = base + (long)srow * ld + ch * 8;
            const u32x4 kv = *reinterpret_cast<const u32x4 *>(src + W);
            u32x4 vv = *reinterpret_cast<const u32x4 *>(src + 2 * W);
            *reinterpret_cast<u32x4 *>(ldsK + row * 128 + ((ch ^ (row & 7)) << 4)) = kv;
            // V: 8-byte slot u -> u ^ ((row>>1)&3): chunk moves by bit 1, halves swap by bit 0
            if ((row >> 1) & 1) vv = u32x4{vv[2], vv[3], vv[0], vv[1]};
            *reinterpret_cast<u32x4 *>(ldsV + row * 128 + ((ch ^ ((row >> 2) & 1)) << 4)) = vv;
        }
    }
    // Query tiles.  A sequence of 16 n + 1 tokens whose tile count leaves ONE tile over after whole rounds of
    // the waves (S = 257 on 8 waves: 17 tiles) would keep seven waves idle for a third round that holds a
    // single query row: that tile's KEYS are split over all waves instead and the partial (m, l, O) merged
    // through LDS (V2 kernels, not causal).  Whether a tile is split depends on S alone, never on q_rows, so
    // ec_attention_rows stays a bit-exact prefix of ec_attention.
    const int n_qt_all = (S + 15) / 16;
    const bool lone = V2 && attn_lone_tile(S, a.causal, AT_WAVES);
    const int n_qt_req = (a.q_rows + 15) / 16;                           // tiles the caller asked for
    const int n_qt = lone ? min(n_qt_req, n_qt_all - 1) : n_qt_req;      // ... that are walked tile by tile
    const bool do_lone = lone && n_qt_req == n_qt_all;
    // first Q tile of this wave, issued before the barrier so its latency hides behind staging
    v8 qf[2], qn[2];
    auto load_q = [&](int qt, v8(&dst)[2]) {
        int qsrc = qt * 16 + c_lane;
        qsrc = qsrc < S ? qsrc : S - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
            dst[ks] = *reinterpret_cast<const v8 *>(base + (long)qsrc * ld + ks * 32 + g_lane * 8);
    };
    if (wave < n_qt)
        load_q(wave, qn);
    else if (do_lone)
        load_q(n_qt_all - 1, qn);
    __syncthreads();
    attn_stamp(1);

    v8 ones;
#pragma unroll
    for (int j = 0; j < 8; j++) ones[j] = to16(1.f, elem());
    asm volatile("" : "+v"(ones));   // stays in four VGPRs (else rebuilt from SGPRs in every block)
    // q <- q * log2(e) / sqrt(64), rounded to the operand type once per tile
    auto scale_q = [&](v8(&q)[2]) {
        if (QM != QM_KERNEL) return;
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int j = 0; j < 8; j++) q[ks][j] = to16((float)q[ks][j] * a.scale_log2e, elem());
    };
    for (int qt = wave; qt < n_qt; qt += AT_WAVES) {
        qf[0] = qn[0], qf[1] = qn[1];
        // prefetch the next tile's Q (after the last one: the split tile's)
        if (qt + AT_WAVES < n_qt)
            load_q(qt + AT_WAVES, qn);
        else if (do_lone)
            load_q(n_qt_all - 1, qn);
        const int qrow = qt * 16 + c_lane;
        // lane coordinates made opaque per tile: otherwise hipcc hoists the per-lane LDS addresses of EVERY
        // block variant out of the tile loop, keeps them live across it and spills (scratch reloads in the
        // tile path cost a memory round trip each)
        int g = g_lane, c16 = c_lane;
        asm volatile("" : "+v"(g), "+v"(c16));
        float lsum, m_log2;   // softmax denominator and maximum (log2 domain, scaled scores)
        f32x4 o[5];
#pragma unroll
        for (int dt = 0; dt < 5; dt++) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (V2) {
            f32x4 mneg = f32x4{0.f, 0.f, 0.f, 0.f};
            scale_q(qf);
            if (!a.causal) {
                attn_keys<DT, QM>(ldsK, ldsV, S, 0, S >> 5, true, qf, ones, a.scale_log2e, mneg, o, g, c16);
            } else {
                const int klimit = qrow < S ? qrow + 1 : S;   // keys < klimit are visible
                // rows of this tile see no key beyond 16 qt + 15: skip the blocks past it; blocks entirely
                // below every lane's klimit (16 qt + 1 ..) need no masking
                const int kend = min(SP, ((qt * 16 + 16 + 31) / 32) * 32);
                const int kfree = min(qt * 16 + 1, kend);
                bool down = true;
                int key0 = 0;
                for (; key0 + 64 <= kfree; key0 += 64, down = false)
                    attn_block2<DT, 2, false, QM>(ldsK, ldsV, key0, klimit, qf, ones, down, a.scale_log2e, mneg, o, g, c16);
                for (; key0 + 64 <= kend; key0 += 64, down = false)
                    attn_block2<DT, 2, true, QM>(ldsK, ldsV, key0, klimit, qf, ones, down, a.scale_log2e, mneg, o, g, c16);
                if (key0 < kend)
                    attn_block2<DT, 1, true, QM>(ldsK, ldsV, key0, klimit, qf, ones, down, a.scale_log2e, mneg, o, g, c16);
            }
            lsum = o[4][0];      // every row of the ones tile carries the query's row sum
            m_log2 = QM != QM_RAW ? -mneg[0] : -mneg[0] * a.scale_log2e;
        } else {
            const int klimit = a.causal ? (qrow < S ? qrow + 1 : S) : S;
            const int kend = a.causal ? min(SP, ((qt * 16 + 16 + 31) / 32) * 32) : SP;
            const int kfree = a.causal ? qt * 16 + 1 : S;
            float m_run = -1e30f, l_run = 0.f;
            f32x4 o4[4];
#pragma unroll
            for (int dt = 0; dt < 4; dt++) o4[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            int key0 = 0;
            for (; key0 + 64 <= kend; key0 += 64) {
                if (key0 + 64 <= kfree)
                    attn_block<DT, 2, false>(ldsK, ldsV, key0, klimit, qf, a.scale_log2e, m_run, l_run, o4,
                                             g, c16);
                else
                    attn_block<DT, 2, true>(ldsK, ldsV, key0, klimit, qf, a.scale_log2e, m_run, l_run, o4,
                                            g, c16);
            }
            if (key0 < kend)
                attn_block<DT, 1, true>(ldsK, ldsV, key0, klimit, qf, a.scale_log2e, m_run, l_run, o4, g,
                                        c16);
            lsum = xor_sum(l_run);
            m_log2 = m_run * a.scale_log2e;
#pragma unroll
            for (int dt = 0; dt < 4; dt++) o[dt] = o4[dt];
        }

        // ---- normalise and store: lane owns query c16, head dims 16 g .. 16 g + 15 ----
        const float inv = 1.f / lsum;
        if (LSE && qrow < a.q_rows && g == 0)
            a.lse[((long)seq * a.heads + head) * S + qrow] = m_log2 + __builtin_amdgcn_logf(lsum);
        if (qrow < a.q_rows) {
            elem ov[16];
#pragma unroll
            for (int dt = 0; dt < 4; dt++)
#pragma unroll
                for (int r = 0; r < 4; r++) ov[4 * dt + r] = to16(o[dt][r] * inv, elem());
            elem *dst = (elem *)a.out + ((long)seq * a.q_rows + qrow) * W + head * 64 + g * 16;
            *reinterpret_cast<u32x4 *>(dst) = *reinterpret_cast<const u32x4 *>(&ov[0]);
            *reinterpret_cast<u32x4 *>(dst + 8) = *reinterpret_cast<const u32x4 *>(&ov[8]);
        }
    }
    if constexpr (V2) {
        if (do_lone) {
            // ---- the tile left over: this wave's share of its keys, then the merge ----
            float *part = reinterpret_cast<float *>(smem + 2 * SP * 128);
            const int steps = S >> 5;
            const int s0 = wave * steps / AT_WAVES, s1 = (wave + 1) * steps / AT_WAVES;
            const bool tail = wave == AT_WAVES - 1;
            int g = g_lane, c16 = c_lane;
            asm volatile("" : "+v"(g), "+v"(c16));
            qf[0] = qn[0], qf[1] = qn[1];
            scale_q(qf);
            f32x4 o[5];
#pragma unroll
            for (int dt = 0; dt < 5; dt++) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 mneg = f32x4{0.f, 0.f, 0.f, 0.f};
            attn_keys<DT, QM>(ldsK, ldsV, S, s0, s1, tail, qf, ones, a.scale_log2e, mneg, o, g, c16);
            if (c16 == 0) {   // the tile's one valid query (row S - 1) lives in lanes 0, 16, 32, 48
                float *dst = part + wave * ATTN_PART;
#pragma unroll
                for (int dt = 0; dt < 4; dt++)
                    *reinterpret_cast<f32x4 *>(dst + 16 * g + 4 * dt) = o[dt];
                if (g == 0) {
                    const float mw = QM != QM_RAW ? -mneg[0] : -mneg[0] * a.scale_log2e;   // log2 units
                    dst[64] = (s1 > s0 || tail) ? mw : -1e30f;   // a wave without keys: weight 0
                    dst[65] = o[4][0];
                }
            }
            __syncthreads();
            if (wave == 0) {   // lane = head dim
                float m = -1e30f;
#pragma unroll
                for (int w = 0; w < AT_WAVES; w++) m = fmaxf(m, part[w * ATTN_PART + 64]);
                float num = 0.f, den = 0.f;
#pragma unroll
                for (int w = 0; w < AT_WAVES; w++) {
                    const float f = __builtin_amdgcn_exp2f(part[w * ATTN_PART + 64] - m);
                    num = __builtin_fmaf(f, part[w * ATTN_PART + lane], num);
                    den = __builtin_fmaf(f, part[w * ATTN_PART + 65], den);
                }
                const int qrow = S - 1;
                ((elem *)a.out)[((long)seq * a.q_rows + qrow) * W + head * 64 + lane] = to16(num / den, elem());
                if (LSE && lane == 0)
                    a.lse[((long)seq * a.heads + head) * S + qrow] = m + __builtin_amdgcn_logf(den);
            }
        }
    }
    attn_stamp(2);
#ifdef EC_ATTN_DIAG
    if (threadIdx.x == 0 && blockIdx.x < 65536) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        ec_attn_stamps[blockIdx.x * 4 + 3] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
}


// ---------------------------------------------------------------------------------------
// Attention on hi + lo fp16 operands (round 5): the fp32-class attention of the tolerance mode's first blocks
// (ec_vit_weights.precise_attn_blocks) on the 16-bit matrix instruction instead of the fp32 one -- 2.5 PFLOP/s against
// 157 TFLOP/s of peak, three products per tile:
//   scores   s = q_hi k_hi + q_lo k_hi + q_hi k_lo            (q = c (q_hi + q_lo) re-split in registers, c = log2 e / 8)
//   P V      o += v_hi p_hi + v_hi p_lo + v_lo p_hi,  l += 1 p_hi + 1 p_lo     (p = exp2(s - m) as p_hi + p_lo)
// (the lo . lo products, 2^-22 of the result, are left out).  Structure of attention_kernel: one 8-wave workgroup per
// (sequence, head), K_hi, K_lo, V_hi, V_lo of the head in LDS (4 x S x 128 B: S <= 288 in 160 KiB -- longer sequences take
// ec_attention_split's fp32 kernel), S^T = K Q^T so that a lane holds one query column, P as the B operand of O^T = V^T P^T,
// the running maximum that moves only when a score exceeds it by 2^10, the odd key of S = 32 n + 1 as a rank-one update (in
// fp32 on the vector ALU), the lone 17th tile of S = 257 split over the waves.  Plain builtin MFMAs and plain C++ for the
// seldom-taken rescale (no inline-asm accumulators: nothing for tools/check_attn_isa.py to check here).
// ---------------------------------------------------------------------------------------
struct AttnHlArgs {
    const _Float16 *qkv_hi, *qkv_lo;   // [n_seq * S, 3W]: q | k | v, a PLAIN q
    _Float16 *out_hi, *out_lo;         // [n_seq * S, W]
    int S, W, heads;
};

__device__ __forceinline__ void hl_move(float d, bool down, f32x4 &mneg, f32x4 (&o)[5])
{
    float alpha = __builtin_amdgcn_exp2f(-d);
    alpha = down ? 0.f : alpha;
#pragma unroll
    for (int dt = 0; dt < 5; dt++) o[dt] *= alpha;
    mneg -= d;
}

// one 32-key step.  `down`: the tile has added nothing yet (m may move down too)
template <bool MASK>
__device__ __forceinline__ void attn_step_hl(const unsigned char *kh, const unsigned char *kl, const unsigned char *vh,
                                             const unsigned char *vl, int key0, int klimit, const f16x8 (&qh)[2],
                                             const f16x8 (&ql)[2], const f16x8 &ones, bool down, f32x4 &mneg, f32x4 (&o)[5],
                                             int g, int c16)
{
    f32x4 acc[2];
#pragma unroll
    for (int kt = 0; kt < 2; kt++) {
        const int row = key0 + kt * 16 + c16;
        const int o0 = row * 128 + (((0 + g) ^ (row & 7)) << 4), o1 = row * 128 + (((4 + g) ^ (row & 7)) << 4);
        const f16x8 k0h = *reinterpret_cast<const f16x8 *>(kh + o0), k1h = *reinterpret_cast<const f16x8 *>(kh + o1);
        const f16x8 k0l = *reinterpret_cast<const f16x8 *>(kl + o0), k1l = *reinterpret_cast<const f16x8 *>(kl + o1);
        f32x4 a = mfma16(k0l, qh[0], mneg);       // the small products first
        a = mfma16(k1l, qh[1], a);
        a = mfma16(k0h, ql[0], a);
        a = mfma16(k1h, ql[1], a);
        a = mfma16(k0h, qh[0], a);
        acc[kt] = mfma16(k1h, qh[1], a);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; kt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (MASK) {
                const int key = key0 + kt * 16 + 4 * g + r;
                acc[kt][r] = key < klimit ? acc[kt][r] : -INFINITY;
            }
            mx = fmaxf(mx, acc[kt][r]);
        }
    if (__builtin_amdgcn_ballot_w64(mx > ATTN_THR || (down && mx < -ATTN_LO)) != 0) {
        const float mq = xor_max(mx);
        const float d = (mq > ATTN_THR || (down && mq < -ATTN_LO)) ? mq : 0.f;      // only the queries that need it move
        acc[0] -= d, acc[1] -= d;
        hl_move(d, down, mneg, o);
    }
    // V^T fragments of both parts: A operand row i <-> head dim (i >> 2) * 16 + 4 dt + (i & 3)
    f16x8 vfh[4], vfl[4];
#pragma unroll
    for (int dt = 0; dt < 4; dt++)
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const int row = key0 + 16 * hh + 4 * g + (c16 >> 2);
            const int off = row * 128 + ((((c16 & 3) * 4 + dt) ^ ((row >> 1) & 3)) << 3);
            const f16x4 th = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(vh + off)));
            const f16x4 tl = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(vl + off)));
#pragma unroll
            for (int e = 0; e < 4; e++) vfh[dt][4 * hh + e] = th[e], vfl[dt][4 * hh + e] = tl[e];
        }
    // P = exp2(s - m) as hi + lo; B operand element j <-> key key0 + 16 (j >> 2) + 4 g + (j & 3)
    f16x8 ph, pl;
#pragma unroll
    for (int kt = 0; kt < 2; kt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float pv = __builtin_amdgcn_exp2f(acc[kt][r]);
            const _Float16 h = (_Float16)pv;
            ph[4 * kt + r] = h;
            pl[4 * kt + r] = (_Float16)(pv - (float)h);
        }
    o[4] = mfma16(ones, pl, o[4]);
    o[4] = mfma16(ones, ph, o[4]);
#pragma unroll
    for (int dt = 0; dt < 4; dt++) {
        f32x4 t = mfma16(vfl[dt], ph, o[dt]);
        t = mfma16(vfh[dt], pl, t);
        o[dt] = mfma16(vfh[dt], ph, t);
    }
}

// the single key behind the last full step (S = 32 n + 1): the LDS rows behind it are copies of it, so every accumulator
// register of a lane is the lane's query against THIS key; P V in fp32 on the vector ALU (no rounding of p at all)
__device__ __forceinline__ void attn_odd_key_hl(const unsigned char *kh, const unsigned char *kl, const unsigned char *vh,
                                                const unsigned char *vl, int key, const f16x8 (&qh)[2], const f16x8 (&ql)[2],
                                                bool down, f32x4 &mneg, f32x4 (&o)[5], int g, int c16)
{
    const int row = key + c16;
    const int o0 = row * 128 + (((0 + g) ^ (row & 7)) << 4), o1 = row * 128 + (((4 + g) ^ (row & 7)) << 4);
    const f16x8 k0h = *reinterpret_cast<const f16x8 *>(kh + o0), k1h = *reinterpret_cast<const f16x8 *>(kh + o1);
    const f16x8 k0l = *reinterpret_cast<const f16x8 *>(kl + o0), k1l = *reinterpret_cast<const f16x8 *>(kl + o1);
    f16x4 vvh[4], vvl[4];
#pragma unroll
    for (int dt = 0; dt < 4; dt++) {      // logical 8-byte slot 4 g + dt of V's row, stored at slot ^ ((key >> 1) & 3)
        const int off = key * 128 + ((4 * g + dt) ^ ((key >> 1) & 3)) * 8;
        vvh[dt] = *reinterpret_cast<const f16x4 *>(vh + off), vvl[dt] = *reinterpret_cast<const f16x4 *>(vl + off);
    }
    f32x4 a = mfma16(k0l, qh[0], mneg);
    a = mfma16(k1l, qh[1], a);
    a = mfma16(k0h, ql[0], a);
    a = mfma16(k1h, ql[1], a);
    a = mfma16(k0h, qh[0], a);
    a = mfma16(k1h, qh[1], a);
    float sc = a[0];
    if (__builtin_amdgcn_ballot_w64(sc > ATTN_THR || (down && sc < -ATTN_LO)) != 0) {
        const float d = (sc > ATTN_THR || (down && sc < -ATTN_LO)) ? sc : 0.f;
        sc -= d;
        hl_move(d, down, mneg, o);
    }
    const float pv = __builtin_amdgcn_exp2f(sc);
    o[4][0] += pv;
#pragma unroll
    for (int dt = 0; dt < 4; dt++)
#pragma unroll
        for (int r = 0; r < 4; r++) o[dt][r] = __builtin_fmaf(pv, (float)vvh[dt][r] + (float)vvl[dt][r], o[dt][r]);
}

// keys of one tile: the full 32-key steps [step0, step1), then (`tail`) what lies behind the last full step
__device__ __forceinline__ void attn_keys_hl(const unsigned char *kh, const unsigned char *kl, const unsigned char *vh,
                                             const unsigned char *vl, int S, int step0, int step1, bool tail,
                                             const f16x8 (&qh)[2], const f16x8 (&ql)[2], const f16x8 &ones, f32x4 &mneg,
                                             f32x4 (&o)[5], int g, int c16, bool down = true)
{
    for (int s = step0; s < step1; s++) {
        attn_step_hl<false>(kh, kl, vh, vl, 32 * s, S, qh, ql, ones, down, mneg, o, g, c16);
        down = false;
    }
    if (tail) {
        const int full = S >> 5, nt = S - 32 * full;
        if (nt == 1)
            attn_odd_key_hl(kh, kl, vh, vl, S - 1, qh, ql, down, mneg, o, g, c16);
        else if (nt > 1)
            attn_step_hl<true>(kh, kl, vh, vl, 32 * full, S, qh, ql, ones, down, mneg, o, g, c16);
    }
}

constexpr int HL_WAVES = 8;
__global__ __launch_bounds__(HL_WAVES * 64, 1) void attention_hl_kernel(const AttnHlArgs a)
{
    constexpr int THREADS = HL_WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int S = a.S, W = a.W;
    const int SP = 32 * ((S + 31) / 32);
    unsigned char *kh = smem, *kl = smem + SP * 128, *vh = smem + 2 * SP * 128, *vl = smem + 3 * SP * 128;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g_lane = lane >> 4, c_lane = lane & 15;
    const int head = blockIdx.x % a.heads, seq = blockIdx.x / a.heads;
    const long ld = 3L * W;
    const long base = (long)seq * S * ld + head * 64;
    {   // stage the four planes: every load of a pass pair in flight before its LDS writes
        const int r_in = threadIdx.x >> 3, ch = threadIdx.x & 7;
        for (int p0 = 0; p0 < SP; p0 += 2 * (THREADS / 8)) {
            u32x4 t[2][4];
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const int row = p0 + r_in + p * (THREADS / 8);
                if (row < SP) {
                    const long src = base + (long)(row < S ? row : S - 1) * ld + ch * 8;   // rows behind the last key: copies of it
                    t[p][0] = *reinterpret_cast<const u32x4 *>(a.qkv_hi + src + W), t[p][1] = *reinterpret_cast<const u32x4 *>(a.qkv_lo + src + W);
                    t[p][2] = *reinterpret_cast<const u32x4 *>(a.qkv_hi + src + 2 * W), t[p][3] = *reinterpret_cast<const u32x4 *>(a.qkv_lo + src + 2 * W);
                }
            }
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const int row = p0 + r_in + p * (THREADS / 8);
                if (row < SP) {
                    const int ko = row * 128 + ((ch ^ (row & 7)) << 4), vo = row * 128 + ((ch ^ ((row >> 2) & 1)) << 4);
                    *reinterpret_cast<u32x4 *>(kh + ko) = t[p][0];
                    *reinterpret_cast<u32x4 *>(kl + ko) = t[p][1];
                    u32x4 x = t[p][2], y = t[p][3];
                    if ((row >> 1) & 1) x = u32x4{x[2], x[3], x[0], x[1]}, y = u32x4{y[2], y[3], y[0], y[1]};
                    *reinterpret_cast<u32x4 *>(vh + vo) = x;
                    *reinterpret_cast<u32x4 *>(vl + vo) = y;
                }
            }
        }
    }
    const int n_qt_all = (S + 15) / 16;
    const bool lone = attn_lone_tile(S, 0, HL_WAVES);
    const int n_qt = lone ? n_qt_all - 1 : n_qt_all;
    constexpr float C = 0.125f * 1.4426950408889634f;
    // q of one tile: c (q_hi + q_lo) in fp32, split again (the softmax scale and the base change enter before the split)
    auto load_q = [&](int qt, f16x8(&qh)[2], f16x8(&ql)[2]) {
        int qsrc = qt * 16 + c_lane;
        qsrc = qsrc < S ? qsrc : S - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const f16x8 h = *reinterpret_cast<const f16x8 *>(a.qkv_hi + base + (long)qsrc * ld + ks * 32 + g_lane * 8);
            const f16x8 l = *reinterpret_cast<const f16x8 *>(a.qkv_lo + base + (long)qsrc * ld + ks * 32 + g_lane * 8);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                float x = ((float)h[j] + (float)l[j]) * C;
                asm volatile("" : "+v"(x));
                qh[ks][j] = (_Float16)x;
                ql[ks][j] = (_Float16)(x - (float)qh[ks][j]);
            }
        }
    };
    f16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; j++) ones[j] = (_Float16)1.f;
    asm volatile("" : "+v"(ones));
    __syncthreads();
    auto store = [&](int qrow, int g, const f32x4(&o)[5], float inv) {
        _Float16 hi[16], lo[16];
#pragma unroll
        for (int dt = 0; dt < 4; dt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float x = o[dt][r] * inv;
                asm volatile("" : "+v"(x));      // ONE rounded fp32 value for both parts (see attention_f32m_kernel)
                hi[4 * dt + r] = (_Float16)x;
                lo[4 * dt + r] = (_Float16)(x - (float)hi[4 * dt + r]);
            }
        const long off = ((long)seq * S + qrow) * W + head * 64 + g * 16;
        *reinterpret_cast<u32x4 *>(a.out_hi + off) = *reinterpret_cast<const u32x4 *>(&hi[0]);
        *reinterpret_cast<u32x4 *>(a.out_hi + off + 8) = *reinterpret_cast<const u32x4 *>(&hi[8]);
        *reinterpret_cast<u32x4 *>(a.out_lo + off) = *reinterpret_cast<const u32x4 *>(&lo[0]);
        *reinterpret_cast<u32x4 *>(a.out_lo + off + 8) = *reinterpret_cast<const u32x4 *>(&lo[8]);
    };
    for (int qt = wave; qt < n_qt; qt += HL_WAVES) {
        f16x8 qh[2], ql[2];
        load_q(qt, qh, ql);
        int g = g_lane, c16 = c_lane;
        asm volatile("" : "+v"(g), "+v"(c16));
        f32x4 o[5];
#pragma unroll
        for (int dt = 0; dt < 5; dt++) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 mneg = f32x4{0.f, 0.f, 0.f, 0.f};
        attn_keys_hl(kh, kl, vh, vl, S, 0, S >> 5, true, qh, ql, ones, mneg, o, g, c16);
        const int qrow = qt * 16 + c16;
        if (qrow < S) store(qrow, g, o, 1.f / o[4][0]);
    }
    if (lone) {
        // the tile left over (one query row): this wave's share of its keys, then the merge through LDS
        float *part = reinterpret_cast<float *>(smem + 4 * SP * 128);
        const int steps = S >> 5, s0 = wave * steps / HL_WAVES, s1 = (wave + 1) * steps / HL_WAVES;
        const bool tail = wave == HL_WAVES - 1;
        f16x8 qh[2], ql[2];
        load_q(n_qt_all - 1, qh, ql);
        int g = g_lane, c16 = c_lane;
        asm volatile("" : "+v"(g), "+v"(c16));
        f32x4 o[5];
#pragma unroll
        for (int dt = 0; dt < 5; dt++) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 mneg = f32x4{0.f, 0.f, 0.f, 0.f};
        attn_keys_hl(kh, kl, vh, vl, S, s0, s1, tail, qh, ql, ones, mneg, o, g, c16);
        if (c16 == 0) {   // the tile's one valid query (row S - 1) lives in lanes 0, 16, 32, 48
            float *dst = part + wave * ATTN_PART;
#pragma unroll
            for (int dt = 0; dt < 4; dt++) *reinterpret_cast<f32x4 *>(dst + 16 * g + 4 * dt) = o[dt];
            if (g == 0) {
                dst[64] = (s1 > s0 || tail) ? -mneg[0] : -1e30f;   // a wave without keys: weight 0
                dst[65] = o[4][0];
            }
        }
        __syncthreads();
        if (wave == 0) {   // lane = head dim
            float m = -1e30f;
#pragma unroll
            for (int w = 0; w < HL_WAVES; w++) m = fmaxf(m, part[w * ATTN_PART + 64]);
            float num = 0.f, den = 0.f;
#pragma unroll
            for (int w = 0; w < HL_WAVES; w++) {
                const float f = __builtin_amdgcn_exp2f(part[w * ATTN_PART + 64] - m);
                num = __builtin_fmaf(f, part[w * ATTN_PART + lane], num);
                den = __builtin_fmaf(f, part[w * ATTN_PART + 65], den);
            }
            float x = num / den;
            asm volatile("" : "+v"(x));
            const _Float16 h = (_Float16)x;
            const long off = ((long)seq * S + S - 1) * W + head * 64 + lane;
            a.out_hi[off] = h;
            a.out_lo[off] = (_Float16)(x - (float)h);
        }
    }
}

// The same for sequences whose four planes do not fit the LDS but half of them does (288 < S <= 608: S = 577): the keys in
// TWO passes of one workgroup -- stage K_hi, K_lo, V_hi, V_lo of keys [0, split), walk every query tile of the wave against
// them, stage the rest, walk the tiles again -- with each tile's (m, l, O) kept in registers between the passes (at most five
// tiles per wave x 24 registers: the 8-wave workgroup has 256 per lane).  No exchange, no merge: a tile's running maximum
// simply goes on where the first pass left it.
constexpr int HL2_TILES = 5;      // query tiles per wave at most: S <= 8 x 5 x 16 = 640
__global__ __launch_bounds__(HL_WAVES * 64, 1) void attention_hl2_kernel(const AttnHlArgs a, int split, int SPL)
{
    constexpr int THREADS = HL_WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int S = a.S, W = a.W;
    unsigned char *kh = smem, *kl = smem + SPL * 128, *vh = smem + 2 * SPL * 128, *vl = smem + 3 * SPL * 128;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g_lane = lane >> 4, c_lane = lane & 15;
    const int head = blockIdx.x % a.heads, seq = blockIdx.x / a.heads;
    const long ld = 3L * W;
    const long base = (long)seq * S * ld + head * 64;
    const int n_qt = (S + 15) / 16;
    constexpr float C = 0.125f * 1.4426950408889634f;
    auto load_q = [&](int qt, f16x8(&qh)[2], f16x8(&ql)[2]) {
        int qsrc = qt * 16 + c_lane;
        qsrc = qsrc < S ? qsrc : S - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const f16x8 h = *reinterpret_cast<const f16x8 *>(a.qkv_hi + base + (long)qsrc * ld + ks * 32 + g_lane * 8);
            const f16x8 l = *reinterpret_cast<const f16x8 *>(a.qkv_lo + base + (long)qsrc * ld + ks * 32 + g_lane * 8);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                float x = ((float)h[j] + (float)l[j]) * C;
                asm volatile("" : "+v"(x));
                qh[ks][j] = (_Float16)x;
                ql[ks][j] = (_Float16)(x - (float)qh[ks][j]);
            }
        }
    };
    f16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; j++) ones[j] = (_Float16)1.f;
    asm volatile("" : "+v"(ones));
    f32x4 o[HL2_TILES][5];
    float mneg[HL2_TILES];          // (-m of a tile: one register between the passes, the MFMA's C operand inside one)
#pragma unroll
    for (int t = 0; t < HL2_TILES; t++) {
        mneg[t] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 5; dt++) o[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int half = 0; half < 2; half++) {
        const int key0 = half ? split : 0, Sl = half ? S - split : split;
        if (half) __syncthreads();      // every wave is done with the first pass's planes
        {
            const int r_in = threadIdx.x >> 3, ch = threadIdx.x & 7;
            for (int p0 = 0; p0 < SPL; p0 += 2 * (THREADS / 8)) {
                u32x4 t[2][4];
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    const int row = p0 + r_in + p * (THREADS / 8);
                    if (row < SPL) {
                        const long src = base + (long)(key0 + (row < Sl ? row : Sl - 1)) * ld + ch * 8;
                        t[p][0] = *reinterpret_cast<const u32x4 *>(a.qkv_hi + src + W), t[p][1] = *reinterpret_cast<const u32x4 *>(a.qkv_lo + src + W);
                        t[p][2] = *reinterpret_cast<const u32x4 *>(a.qkv_hi + src + 2 * W), t[p][3] = *reinterpret_cast<const u32x4 *>(a.qkv_lo + src + 2 * W);
                    }
                }
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    const int row = p0 + r_in + p * (THREADS / 8);
                    if (row < SPL) {
                        const int ko = row * 128 + ((ch ^ (row & 7)) << 4), vo = row * 128 + ((ch ^ ((row >> 2) & 1)) << 4);
                        *reinterpret_cast<u32x4 *>(kh + ko) = t[p][0];
                        *reinterpret_cast<u32x4 *>(kl + ko) = t[p][1];
                        u32x4 x = t[p][2], y = t[p][3];
                        if ((row >> 1) & 1) x = u32x4{x[2], x[3], x[0], x[1]}, y = u32x4{y[2], y[3], y[0], y[1]};
                        *reinterpret_cast<u32x4 *>(vh + vo) = x;
                        *reinterpret_cast<u32x4 *>(vl + vo) = y;
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < HL2_TILES; t++) {
            const int qt = wave + HL_WAVES * t;
            if (qt < n_qt) {      // (wave-uniform)
                f16x8 qh[2], ql[2];
                load_q(qt, qh, ql);
                int g = g_lane, c16 = c_lane;
                asm volatile("" : "+v"(g), "+v"(c16));
                f32x4 mn = f32x4{mneg[t], mneg[t], mneg[t], mneg[t]};
                attn_keys_hl(kh, kl, vh, vl, Sl, 0, Sl >> 5, true, qh, ql, ones, mn, o[t], g, c16, half == 0);
                mneg[t] = mn[0];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < HL2_TILES; t++) {
        const int qt = wave + HL_WAVES * t, qrow = qt * 16 + c_lane;
        if (qt < n_qt && qrow < S) {
            const float inv = 1.f / o[t][4][0];
            _Float16 hi[16], lo[16];
#pragma unroll
            for (int dt = 0; dt < 4; dt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float x = o[t][dt][r] * inv;
                    asm volatile("" : "+v"(x));
                    hi[4 * dt + r] = (_Float16)x;
                    lo[4 * dt + r] = (_Float16)(x - (float)hi[4 * dt + r]);
                }
            const long off = ((long)seq * S + qrow) * W + head * 64 + g_lane * 16;
            *reinterpret_cast<u32x4 *>(a.out_hi + off) = *reinterpret_cast<const u32x4 *>(&hi[0]);
            *reinterpret_cast<u32x4 *>(a.out_hi + off + 8) = *reinterpret_cast<const u32x4 *>(&hi[8]);
            *reinterpret_cast<u32x4 *>(a.out_lo + off) = *reinterpret_cast<const u32x4 *>(&lo[0]);
            *reinterpret_cast<u32x4 *>(a.out_lo + off + 8) = *reinterpret_cast<const u32x4 *>(&lo[8]);
        }
    }
}

#ifdef EC_ATTN_DIAG
// ---------------------------------------------------------------------------------------
// DIAGNOSTIC BUILD ONLY (measured, not kept: profiles/r5_attention.md).
// Key-half workgroup pair (round 5) for sequences whose K and V leave room for one 16-wave workgroup per CU only
// (S = 577: 148 KiB).  Each (sequence, head) is handled by TWO 8-wave workgroups that stage HALF of the keys and
// values each (keys [0, split) / [split, S), split a multiple of 32), so two workgroups are resident per CU and one's
// staging runs under the other's compute (the one-workgroup kernel spends 26 % of its time in un-overlapped staging,
// profiles/r4_attention.md) and the 37 query tiles make 4.6 rounds of 8 waves instead of 2.3 of 16 (92 % against 77 %
// occupancy).  Every workgroup walks ALL query tiles against its keys; a tile's partial (m, l, O) goes to a fp32
// workspace, a ticket per (unit, tile) says who is second, and the second one merges -- always as (half 0, half 1), so
// the result does not depend on who finished first -- and writes the output.  The pair is mapped onto ONE XCD
// (workgroup ids b and b + 8 share an XCD): the exchange stays in that XCD's L2.
// ---------------------------------------------------------------------------------------
constexpr int PAIR_REC = 68;     // floats per query in a partial record: O[64], m (log2 units), l, 2 pad
template <int DT, int QM>
__global__ __launch_bounds__(512, 2) void attention_pair_kernel(const AttnArgs a, float *part, int *ticket, int split, int SPL,
                                                                int n_units)
{
    typedef typename T16<DT>::elem elem;
    typedef typename T16<DT>::v8 v8;
    constexpr int WAVES = 8, THREADS = 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int S = a.S, W = a.W;
    const int w = blockIdx.x, half = (w >> 3) & 1, unit = (w >> 4) * 8 + (w & 7);
    if (unit >= n_units) return;
    const int key0 = half ? split : 0, Sl = half ? S - split : split;      // this workgroup's keys
    unsigned char *ldsK = smem;
    unsigned char *ldsV = smem + SPL * 128;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g_lane = lane >> 4, c_lane = lane & 15;
    const int head = unit % a.heads, seq = unit / a.heads;
    const long ld = 3L * W;
    const elem *base = (const elem *)a.qkv + (long)seq * S * ld + head * 64;
    {   // stage this half's K and V: every load in flight before the first LDS write (rows behind the last key: copies of it)
        const int r_in = threadIdx.x >> 3, ch = threadIdx.x & 7;
        u32x4 kv[5], vv[5];
#pragma unroll
        for (int p = 0; p < 5; p++) {
            const int row = r_in + p * (THREADS / 8);
            if (row < SPL) {
                const int srow = key0 + (row < Sl ? row : Sl - 1);
                const elem *src = base + (long)srow * ld + ch * 8;
                kv[p] = *reinterpret_cast<const u32x4 *>(src + W);
                vv[p] = *reinterpret_cast<const u32x4 *>(src + 2 * W);
            }
        }
#pragma unroll
        for (int p = 0; p < 5; p++) {
            const int row = r_in + p * (THREADS / 8);
            if (row < SPL) {
                *reinterpret_cast<u32x4 *>(ldsK + row * 128 + ((ch ^ (row & 7)) << 4)) = kv[p];
                u32x4 t = vv[p];
                if ((row >> 1) & 1) t = u32x4{t[2], t[3], t[0], t[1]};
                *reinterpret_cast<u32x4 *>(ldsV + row * 128 + ((ch ^ ((row >> 2) & 1)) << 4)) = t;
            }
        }
    }
    const int n_qt_all = (S + 15) / 16, n_qt = (a.q_rows + 15) / 16;
    v8 qf[2], qn[2];
    auto load_q = [&](int qt, v8(&dst)[2]) {
        int qsrc = qt * 16 + c_lane;
        qsrc = qsrc < S ? qsrc : S - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
            dst[ks] = *reinterpret_cast<const v8 *>(base + (long)qsrc * ld + ks * 32 + g_lane * 8);
    };
    if (wave < n_qt) load_q(wave, qn);
    __syncthreads();
    v8 ones;
#pragma unroll
    for (int j = 0; j < 8; j++) ones[j] = to16(1.f, elem());
    asm volatile("" : "+v"(ones));
    for (int qt = wave; qt < n_qt; qt += WAVES) {
        qf[0] = qn[0], qf[1] = qn[1];
        if (qt + WAVES < n_qt) load_q(qt + WAVES, qn);
        int g = g_lane, c16 = c_lane;
        asm volatile("" : "+v"(g), "+v"(c16));
        f32x4 o[5];
#pragma unroll
        for (int dt = 0; dt < 5; dt++) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 mneg = f32x4{0.f, 0.f, 0.f, 0.f};
        if (QM == QM_KERNEL) {
#pragma unroll
            for (int ks = 0; ks < 2; ks++)
#pragma unroll
                for (int j = 0; j < 8; j++) qf[ks][j] = to16((float)qf[ks][j] * a.scale_log2e, elem());
        }
        attn_keys<DT, QM>(ldsK, ldsV, Sl, 0, Sl >> 5, true, qf, ones, a.scale_log2e, mneg, o, g, c16);
        const float l_me = o[4][0];
        const float m_me = QM != QM_RAW ? -mneg[0] : -mneg[0] * a.scale_log2e;     // log2 units
        // ---- this half's partial -> workspace (no wait here: the exchange of all of the wave's tiles comes behind the loop) ----
        // The exchange is made of agent-scope ACCESSES, not fences: stores and loads with the sc1 bit are coherent at the
        // device's coherence point line by line.  (The first version took an acq_rel ticket per tile: hipcc brackets that
        // with buffer_wbl2 / buffer_inv -- a write-back and an invalidate of the whole L2 -- 26 x slower than the kernel it
        // was meant to beat; the second waited per tile for store acknowledgement, ticket and loads in a row: 3.3 x slower.)
        float *mine = part + ((((long)unit * n_qt_all + qt) * 2 + half) * 16 + c16) * PAIR_REC;
        if (a.causal == 2) continue;      // (timing experiment: no exchange at all -- nothing is written)
        if (a.causal != 1) {              // write-back stores: the pair shares an XCD, hence an L2 (the form that is correct and least slow)
#pragma unroll
            for (int dt = 0; dt < 4; dt++) *reinterpret_cast<f32x4 *>(mine + 16 * g + 4 * dt) = o[dt];
            if (g == 0) *reinterpret_cast<float2 *>(mine + 64) = make_float2(m_me, l_me);
            continue;
        }
        // (a.causal == 1: agent-scope write-through stores -- slower still, and the merger's sc1 loads of its OWN partial
        // came back stale in this form)
#pragma unroll
        for (int dt = 0; dt < 4; dt++)
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(mine + 16 * g + 4 * dt), "v"(o[dt]) : "memory");
        if (g == 0) {
            const float2 mlv = make_float2(m_me, l_me);
            asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(mine + 64), "v"(mlv) : "memory");
        }
    }
    if (a.causal == 2) return;
    // ---- exchange: every partial of this wave is acknowledged, then one ticket per tile (lane i takes tile wave + 8 i),
    // then the tiles this workgroup came second on are merged -- as (half 0, half 1) whoever merges -- and written ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int my_tiles = n_qt > wave ? (n_qt - wave + WAVES - 1) / WAVES : 0;
    int old = 0;
    if (lane < my_tiles)
        old = __hip_atomic_fetch_add(ticket + (long)unit * n_qt_all + wave + WAVES * lane, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long second = __builtin_amdgcn_ballot_w64(old != 0);
    for (int i = 0; i < my_tiles; i++) {
        if (!((second >> i) & 1)) continue;       // first on this tile: the other half's workgroup finishes it
        const int qt = wave + WAVES * i, qrow = qt * 16 + c_lane, g = g_lane, c16 = c_lane;
        const long rec = ((long)unit * n_qt_all + qt) * 2;
        const float *p0 = part + ((rec + 0) * 16 + c16) * PAIR_REC, *p1 = part + ((rec + 1) * 16 + c16) * PAIR_REC;
        f32x4 o0[4], o1[4];
        float2 ml0, ml1;
#pragma unroll
        for (int dt = 0; dt < 4; dt++) {
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(o0[dt]) : "v"(p0 + 16 * g + 4 * dt) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(o1[dt]) : "v"(p1 + 16 * g + 4 * dt) : "memory");
        }
        asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(ml0) : "v"(p0 + 64) : "memory");
        asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(ml1) : "v"(p1 + 64) : "memory");
        asm volatile("" : "+v"(o0[0]), "+v"(o0[1]), "+v"(o0[2]), "+v"(o0[3]), "+v"(o1[0]), "+v"(o1[1]), "+v"(o1[2]), "+v"(o1[3]), "+v"(ml0));
        const float m = fmaxf(ml0.x, ml1.x);
        const float f0 = __builtin_amdgcn_exp2f(ml0.x - m), f1 = __builtin_amdgcn_exp2f(ml1.x - m);
        const float inv = 1.f / (ml0.y * f0 + ml1.y * f1);
        if (lane == 0) ticket[(long)unit * n_qt_all + qt] = 0;     // ready for the next launch on this workspace
        if (qrow < a.q_rows) {
            elem ov[16];
#pragma unroll
            for (int dt = 0; dt < 4; dt++)
#pragma unroll
                for (int r = 0; r < 4; r++) ov[4 * dt + r] = to16((o0[dt][r] * f0 + o1[dt][r] * f1) * inv, elem());
            elem *dst = (elem *)a.out + ((long)seq * a.q_rows + qrow) * W + head * 64 + g * 16;
            *reinterpret_cast<u32x4 *>(dst) = *reinterpret_cast<const u32x4 *>(&ov[0]);
            *reinterpret_cast<u32x4 *>(dst + 8) = *reinterpret_cast<const u32x4 *>(&ov[8]);
        }
    }
}

#endif

#ifdef EC_ATTN_DIAG
#include "attention_diag.inc"     // attention32_kernel, the round-1 fp32 kernel: A / B forms of the diagnostic build
#endif

#ifdef EC_ATTN_DIAG
int g_attn_variant = 0;
#endif

template <int DT> int dispatch(const AttnArgs &a, int n_seq, int heads, hipStream_t s)
{
    const int n32 = (a.S + 31) / 32;
    // K and V images, plus the merge area of a tile whose keys are split over the waves -- reserved only where the
    // kernel can take that path for this S (its `lone` condition: not causal, S = 16 n + 1, the tile count one more
    // than a multiple of the wave count), so that S = 609 .. 640 (n32 = 20: exactly 160 KiB of K and V) still fits
    const int kv = 32 * n32 * 128 * 2;
    const int waves_if = kv + 16 * ATTN_PART * 4 > 80 * 1024 ? 16 : 8;      // the wave count the full carve would pick
    const bool may_split = attn_lone_tile(a.S, a.causal, waves_if);
    const int lds = kv + (may_split ? 16 * ATTN_PART * 4 : 0);
    if (lds > 160 * 1024)
        return ec::fail(EC_ERR_UNSUPPORTED, "ec_attention: sequence length %d needs %d bytes of LDS (K and V images%s), "
                        "the CU has 163840: S <= 640%s", a.S, lds, may_split ? " + the split-tile merge area" : "",
                        may_split ? " (608 for this S, whose last query tile is split over the waves)" : "");
    // Waves per workgroup: 16 when K + V leave room for one workgroup per CU only, else 8.  (9..12 waves,
    // which would spread the 17 query tiles of S = 257 over two even passes, measure 0.22 ms against
    // 0.17 ms for 8: the second workgroup no longer co-resides.)
    const bool wide = kv + 16 * ATTN_PART * 4 > 80 * 1024;
    void (*kern)(const AttnArgs) = a.lse ? (wide ? attention_kernel<DT, 16, true> : attention_kernel<DT, 8, true>)
                                         : (wide ? attention_kernel<DT, 16> : attention_kernel<DT, 8>);
    if (a.q_scaled == 1)   // the inference towers (never with a log-sum-exp)
        kern = wide ? attention_kernel<DT, 16, false, true, QM_INPUT> : attention_kernel<DT, 8, false, true, QM_INPUT>;
    if (a.q_scaled == 2)   // a plain q, every score scaled in fp32: q is rounded ONCE (the split-operand blocks behind the
                           // fp32-attention ones; QM_KERNEL's second rounding of q cost configs[2] 30 % there)
        kern = wide ? attention_kernel<DT, 16, false, true, QM_RAW> : attention_kernel<DT, 8, false, true, QM_RAW>;
#ifdef EC_ATTN_DIAG
    if (g_attn_variant == 5 && a.q_scaled && DT == EC_F16 && !a.causal && !a.lse)   // round 4: 32-query tiles (A / B)
        kern = wide ? attention32_kernel<16> : attention32_kernel<8>;
    if (g_attn_variant == 1 && !a.q_scaled)   // round 1 / 2 block (per-block maximum, vector-ALU row sum), for A/B
        kern = a.lse ? (wide ? attention_kernel<DT, 16, true, false> : attention_kernel<DT, 8, true, false>)
                     : (wide ? attention_kernel<DT, 16, false, false> : attention_kernel<DT, 8, false, false>);
#endif
    const int nw = wide ? 16 : 8;
    // always the full carve (one attribute call per kernel and device, thread-safe)
    if (int rc = ec::ensure_dynamic_lds(reinterpret_cast<const void *>(kern), 160 * 1024)) return rc;
    // algorithmic work: QK^T and PV, 2 * 2 * S^2 * 64 flops per head; bytes: read qkv, write out
    ec::ProfScope prof(ec::PROF_ATTENTION, s, 4.0 * a.q_rows * a.S * 64.0 * heads * n_seq,
                       (double)n_seq * a.W * 2.0 * (2.0 * a.S + 2.0 * a.q_rows));
    hipLaunchKernelGGL(kern, dim3((unsigned)heads * (unsigned)n_seq), dim3(nw * 64), lds, s, a);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}


// ---------------------------------------------------------------------------------------
// fp32 attention for the split-precision towers on the fp32 matrix instruction (round 4): v_mfma_f32_16x16x4_f32 takes fp32 operands and is bit for bit an
// fmaf chain (cdna_hip_programming.md, FP32-input MFMA), at the vector ALU's peak rate but with the operands read
// once per 16 x 16 tile instead of once per product.  One workgroup = 64 queries of one (sequence, head), one
// 16-query tile per wave, keys in chunks of 32 through two LDS buffers (the next chunk's rows are in flight in
// registers during the chunk's MFMAs; one barrier per chunk), online softmax in fp32 (e^t through v_exp_f32, see exp_e).
//  * S^T = K . Q^T: A = the chunk's key rows (lane: key l & 15, head dims 16 c + 4 (l >> 4) + e), B = the wave's
//    queries in the same dim order (held in registers, pre-scaled by 1/8: exact), so a lane's accumulators are four
//    keys (4 (l >> 4) + r of each 16-key tile) of ONE query (l & 15);
//  * O^T += V^T . P^T: step (t, r) takes the keys 16 t + 4 g + r (g = 0 .. 3), whose P values are exactly register r of
//    tile t in lane group g: no lane movement; A = V read down its columns (one float per lane and step).
// LDS rows are 64 floats with the 16-byte chunk index XORed with row & 15: the 16 key rows of a b128 read and the
// column reads of V are spread over the banks.  2560 sequences x 16 heads x S = 257: 96 ms -> see profiles/r4_attention.md.
// ---------------------------------------------------------------------------------------
// SIN (round 5, the split-operand blocks of ec_vit_weights.precise_attn_blocks): q | k | v arrive as hi + lo 16-bit parts
// (two [rows, 3W] tensors, what EC_EPI_STORE16 leaves with args.aux) and are joined to fp32 on the way in.  PRE: the q
// columns already hold q log2(e) / sqrt(64) (ec_vit_weights.q_scaled), so the scores are the exponent's arguments in
// log2 units and e^t is one v_exp_f32.
constexpr int F32_KCH = 32;
template <int DT, bool SIN = false, bool PRE = false>
__global__ __launch_bounds__(256) void attention_f32m_kernel(const void *qkv_v, const void *qkv_lo_v, void *out_hi, void *out_lo,
                                                             int S, int W, int heads, int causal)
{
    typedef typename T16<DT>::elem elem;
    typedef typename T16<DT>::v4 v4;
    typedef typename std::conditional<SIN, elem, float>::type in_t;
    const in_t *qkv = static_cast<const in_t *>(qkv_v), *qkv_lo = static_cast<const in_t *>(qkv_lo_v);
    // four consecutive values of q | k | v at element offset `off` as fp32 (SIN: hi + lo)
    auto ld4 = [&](long off) -> f32x4 {
        if constexpr (SIN) {
            const v4 h = *reinterpret_cast<const v4 *>(qkv + off), l = *reinterpret_cast<const v4 *>(qkv_lo + off);
            return f32x4{(float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2], (float)h[3] + (float)l[3]};
        } else {
            return *reinterpret_cast<const f32x4 *>(qkv + off);
        }
    };
    __shared__ __attribute__((aligned(16))) float lk[2][F32_KCH * 64];
    __shared__ __attribute__((aligned(16))) float lv[2][F32_KCH * 64];
    const int n_qb = (S + 63) / 64;
    const int qb = blockIdx.x % n_qb, head = (blockIdx.x / n_qb) % heads, seq = blockIdx.x / (n_qb * heads);
    const long ld = 3L * W;
    const long base = (long)seq * S * ld + head * 64;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, j = lane & 15, g = lane >> 4;
    const int q0 = qb * 64 + wave * 16;
    const bool active = q0 < S;                    // wave-uniform: the tile has a query that exists
    const int query = q0 + j, qsrc = query < S ? query : S - 1;
    f32x4 qf[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const f32x4 v = ld4(base + (long)qsrc * ld + 16 * c + 4 * g);
        qf[c] = PRE ? v : v * 0.125f;
    }
    // keys this workgroup needs: all of them, or (causal) those up to its last query
    const int s_eff = causal ? (qb * 64 + 64 < S ? qb * 64 + 64 : S) : S;
    const int nch = (s_eff + F32_KCH - 1) / F32_KCH;
    // staging: thread -> (key row, 16-byte chunk) pairs idx = t, t + 256 of the chunk's 32 x 16
    f32x4 rk[2], rv[2];
    auto fetch = [&](int ch) {
#pragma unroll
        for (int n = 0; n < 2; n++) {
            const int idx = t + 256 * n, key = ch * F32_KCH + (idx >> 4);
            const int ks = key < S ? key : S - 1;
            const long src = base + (long)ks * ld + (idx & 15) * 4;
            rk[n] = ld4(src + W);
            rv[n] = ld4(src + 2 * W);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int n = 0; n < 2; n++) {
            const int idx = t + 256 * n, row = idx >> 4, chunk = (idx & 15) ^ (row & 15);
            *reinterpret_cast<f32x4 *>(&lk[buf][row * 64 + chunk * 4]) = rk[n];
            *reinterpret_cast<f32x4 *>(&lv[buf][row * 64 + chunk * 4]) = rv[n];
        }
    };
    float m = -INFINITY, l = 0.f;
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; dt++) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // e^t as 2^(t log2 e) on v_exp_f32 (1 ulp), the product with log2 e = hi + lo carried to ~2^-48 relative (one fma):
    // three vector instructions and one transcendental instead of expf's twenty, the same accuracy class
    auto exp_e = [](float t) {
        if constexpr (PRE) return __builtin_amdgcn_exp2f(t);
        return __builtin_amdgcn_exp2f(__builtin_fmaf(t, 1.4426950216293335f, t * 1.92596298909109e-8f));
    };
    fetch(0);
    stash(0);
    __syncthreads();
    for (int ch = 0; ch < nch; ch++) {
        const int buf = ch & 1;
        if (ch + 1 < nch) fetch(ch + 1);
        if (active) {
            const float *K = lk[buf], *V = lv[buf];
            f32x4 sc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                f32x4 kf[2];
#pragma unroll
                for (int tt = 0; tt < 2; tt++) {
                    const int row = tt * 16 + j;
                    kf[tt] = *reinterpret_cast<const f32x4 *>(K + row * 64 + (((4 * c + g) ^ (row & 15)) << 2));
                }
#pragma unroll
                for (int e = 0; e < 4; e++)
#pragma unroll
                    for (int tt = 0; tt < 2; tt++)
                        sc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[tt][e], qf[c][e], sc[tt], 0, 0, 0);
            }
            float mx = -INFINITY;
#pragma unroll
            for (int tt = 0; tt < 2; tt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int key = ch * F32_KCH + tt * 16 + 4 * g + r;
                    const bool ok = key < S && (!causal || key <= query);
                    sc[tt][r] = ok ? sc[tt][r] : -INFINITY;
                    mx = fmaxf(mx, sc[tt][r]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mn = fmaxf(m, mx);
            // nothing of this query's row seen so far (mn = -inf: only rows that do not exist or a causal row whose
            // keys all lie ahead, which cannot happen from chunk 0 on): keep everything at zero
            const float alpha = mn == -INFINITY ? 0.f : exp_e(m - mn);
            const float sub = mn == -INFINITY ? 0.f : mn;
            float ps = 0.f;
#pragma unroll
            for (int tt = 0; tt < 2; tt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    sc[tt][r] = exp_e(sc[tt][r] - sub);
                    ps += sc[tt][r];
                }
            l = l * alpha + ps;
            m = mn;
#pragma unroll
            for (int dt = 0; dt < 4; dt++) o[dt] *= alpha;
#pragma unroll
            for (int tt = 0; tt < 2; tt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = tt * 16 + 4 * g + r;
#pragma unroll
                    for (int dt = 0; dt < 4; dt++) {
                        const float a = V[row * 64 + (((4 * dt + (j >> 2)) ^ (row & 15)) << 2) + (j & 3)];
                        o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, sc[tt][r], o[dt], 0, 0, 0);
                    }
                }
        }
        if (ch + 1 < nch) stash(buf ^ 1);
        __syncthreads();
    }
    if (active) {
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        if (query < S) {
            const float inv = 1.f / l;
#pragma unroll
            for (int dt = 0; dt < 4; dt++) {
                v4 hi, lo;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    // the product as ONE rounded fp32 value for both parts: with -ffp-contract=fast hipcc otherwise
                    // rounds hi from fl(o inv) but takes lo against a hi of its own rounded straight from the exact
                    // product (v_fma_mixlo_f16), and where the two differ (a near-tie) lo comes out with the wrong
                    // sign: one element in ~2000 off by a 16-bit ulp
                    float x = o[dt][r] * inv;
                    asm volatile("" : "+v"(x));
                    hi[r] = to16(x, elem());
                    lo[r] = to16(x - (float)hi[r], elem());
                }
                const long off = ((long)seq * S + query) * W + head * 64 + 16 * dt + 4 * g;
                *reinterpret_cast<v4 *>((elem *)out_hi + off) = hi;
                *reinterpret_cast<v4 *>((elem *)out_lo + off) = lo;
            }
        }
    }
}

}  // namespace

#ifdef EC_ATTN_DIAG
// diagnostic build only (not part of the public header): copy the stamp records to the host
extern "C" __attribute__((visibility("default"))) int ec_attn_stamps_read(unsigned long long *host, int n)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(ec_attn_stamps), (size_t)n * 8) == hipSuccess ? 0 : 1;
}
extern "C" __attribute__((visibility("default"))) void ec_attn_set_variant(int v) { g_attn_variant = v; }
#endif

static int attention_rows(const void *qkv, void *out, int n_seq, int S, int width, int heads, int causal, int q_rows,
                          int q_scaled, int dtype, ec_stream_t stream)
{
    AttnArgs a;
    a.qkv = qkv, a.out = out, a.S = S, a.W = width, a.heads = heads, a.causal = causal;
    a.q_rows = q_rows;
    a.lse = nullptr;
    a.scale_log2e = 0.125f * 1.4426950408889634f;
    a.q_scaled = q_scaled;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == EC_F16) return dispatch<EC_F16>(a, n_seq, heads, s);
    if (dtype == EC_BF16) return dispatch<EC_BF16>(a, n_seq, heads, s);
    return ec::fail(EC_ERR_INVALID, "ec_attention: unknown dtype %d", dtype);
}

// tower_ops.h: a plain q whose scores are scaled in fp32 (no second rounding of q), no mask
int ec_tower::attention_exact_scale(const void *qkv, void *out, int n_seq, int S, int width, int heads, int dtype,
                                    ec_stream_t stream)
{
    EC_REQUIRE(n_seq >= 0 && S > 0 && heads > 0 && width == heads * 64, "attention_exact_scale: bad shape");
    if (n_seq == 0) return EC_OK;
    EC_REQUIRE(qkv && out, "attention_exact_scale: null buffer");
    return attention_rows(qkv, out, n_seq, S, width, heads, 0, S, 2, dtype, stream);
}

extern "C" EC_API int ec_attention(const void *qkv, void *out, int n_seq, int S, int width,
                                   int heads, int causal, int dtype, ec_stream_t stream)
{
    return ec_attention_rows(qkv, out, n_seq, S, width, heads, causal, S, dtype, stream);
}

extern "C" EC_API int ec_attention_rows(const void *qkv, void *out, int n_seq, int S, int width,
                                        int heads, int causal, int q_rows, int dtype,
                                        ec_stream_t stream)
{
    EC_REQUIRE(n_seq >= 0 && S > 0 && heads > 0, "ec_attention: bad shape");
    EC_REQUIRE(q_rows >= 1 && q_rows <= S, "ec_attention: q_rows=%d outside 1..%d", q_rows, S);
    EC_REQUIRE(width == heads * 64, "ec_attention: head dim must be 64 (width %d, heads %d)", width,
               heads);
    if (n_seq == 0) return EC_OK;
    EC_REQUIRE(qkv && out, "ec_attention: null buffer");
    return attention_rows(qkv, out, n_seq, S, width, heads, causal, q_rows, 0, dtype, stream);
}

extern "C" EC_API int ec_attention_scaled_q(const void *qkv, void *out, int n_seq, int S, int width, int heads,
                                            int causal, int q_rows, int dtype, ec_stream_t stream)
{
    EC_REQUIRE(n_seq >= 0 && S > 0 && heads > 0, "ec_attention_scaled_q: bad shape");
    EC_REQUIRE(q_rows >= 1 && q_rows <= S, "ec_attention_scaled_q: q_rows=%d outside 1..%d", q_rows, S);
    EC_REQUIRE(width == heads * 64, "ec_attention_scaled_q: head dim must be 64 (width %d, heads %d)", width, heads);
    if (n_seq == 0) return EC_OK;
    EC_REQUIRE(qkv && out, "ec_attention_scaled_q: null buffer");
    return attention_rows(qkv, out, n_seq, S, width, heads, causal, q_rows, 1, dtype, stream);
}

extern "C" EC_API int ec_attention_train(const void *qkv, void *out, float *lse, int n_seq, int S, int width,
                                         int heads, int dtype, ec_stream_t stream)
{
    EC_REQUIRE(n_seq >= 0 && S > 0 && heads > 0, "ec_attention_train: bad shape");
    EC_REQUIRE(width == heads * 64, "ec_attention_train: head dim must be 64 (width %d, heads %d)", width, heads);
    if (n_seq == 0) return EC_OK;
    EC_REQUIRE(qkv && out && lse, "ec_attention_train: null buffer");
    AttnArgs a;
    a.qkv = qkv, a.out = out, a.S = S, a.W = width, a.heads = heads, a.causal = 0;
    a.q_rows = S;
    a.lse = lse;
    a.scale_log2e = 0.125f * 1.4426950408889634f;
    a.q_scaled = 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == EC_F16) return dispatch<EC_F16>(a, n_seq, heads, s);
    if (dtype == EC_BF16) return dispatch<EC_BF16>(a, n_seq, heads, s);
    return ec::fail(EC_ERR_INVALID, "ec_attention_train: unknown dtype %d", dtype);
}

extern "C" EC_API int ec_attention_f32(const float *qkv, void *out_hi, void *out_lo, int n_seq, int S,
                                       int width, int heads, int causal, int dtype,
                                       ec_stream_t stream)
{
    EC_REQUIRE(n_seq >= 0 && S > 0 && heads > 0, "ec_attention_f32: bad shape");
    EC_REQUIRE(width == heads * 64, "ec_attention_f32: head dim must be 64");
    if (n_seq == 0) return EC_OK;
    EC_REQUIRE(qkv && out_hi && out_lo, "ec_attention_f32: null buffer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ec::ProfScope prof(ec::PROF_ATTENTION, s, 4.0 * S * S * 64.0 * heads * n_seq, 0);
#ifdef EC_ATTN_DIAG
    if (g_attn_variant == 6) {      // the round-1 kernel (vector ALU, K / V rows straight from L2), for the A / B
        const int lds = (16 * 64 + 16 * S) * 4;
        EC_REQUIRE(lds <= 64 * 1024, "ec_attention_f32: sequence length %d too long", S);
        const unsigned grid = (unsigned)n_seq * heads * ((S + 15) / 16);
        if (dtype == EC_F16)
            hipLaunchKernelGGL(attention_f32_kernel<EC_F16>, dim3(grid), dim3(256), lds, s, qkv, out_hi, out_lo, S, width,
                               heads, causal);
        else
            hipLaunchKernelGGL(attention_f32_kernel<EC_BF16>, dim3(grid), dim3(256), lds, s, qkv, out_hi, out_lo, S, width,
                               heads, causal);
        EC_CHECK_HIP(hipGetLastError());
        return EC_OK;
    }
#endif
    const long blocks = (long)n_seq * heads * ((S + 63) / 64);
    EC_REQUIRE(blocks < (1L << 31), "ec_attention_f32: %ld workgroups", blocks);
    if (dtype == EC_F16)
        hipLaunchKernelGGL((attention_f32m_kernel<EC_F16, false, false>), dim3((unsigned)blocks), dim3(256), 0, s, qkv, nullptr, out_hi,
                           out_lo, S, width, heads, causal);
    else if (dtype == EC_BF16)
        hipLaunchKernelGGL((attention_f32m_kernel<EC_BF16, false, false>), dim3((unsigned)blocks), dim3(256), 0, s, qkv, nullptr, out_hi,
                           out_lo, S, width, heads, causal);
    else
        return ec::fail(EC_ERR_INVALID, "ec_attention_f32: unknown dtype %d", dtype);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

// EC_ATTN_SPLIT_F32 (tools/bench_attn_split.py, debug_attn_hl.py: force the fp32-MFMA kernel) is read by the diagnostic
// build only: the product library's kernel choice -- numerics and speed of the tolerance mode -- never depends on the
// process environment
#ifdef EC_ATTN_DIAG
static bool split_force_f32() { return getenv("EC_ATTN_SPLIT_F32") != nullptr; }
#else
static constexpr bool split_force_f32() { return false; }
#endif

extern "C" EC_API int ec_attention_split(const void *qkv_hi, const void *qkv_lo, void *out_hi, void *out_lo, int n_seq, int S,
                                         int width, int heads, int q_prescaled, int dtype, ec_stream_t stream)
{
    EC_REQUIRE(n_seq >= 0 && S > 0 && heads > 0, "ec_attention_split: bad shape");
    EC_REQUIRE(width == heads * 64, "ec_attention_split: head dim must be 64");
    if (n_seq == 0) return EC_OK;
    EC_REQUIRE(qkv_hi && qkv_lo && out_hi && out_lo, "ec_attention_split: null buffer");
    EC_REQUIRE((((uintptr_t)qkv_hi | (uintptr_t)qkv_lo | (uintptr_t)out_hi | (uintptr_t)out_lo) & 15) == 0,
               "ec_attention_split: buffers must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ec::ProfScope prof(ec::PROF_ATTENTION, s, 4.0 * S * S * 64.0 * heads * n_seq, (double)n_seq * S * width * 2.0 * 8.0);
    // a plain f16 q and a sequence whose four planes fit the CU's LDS: three products per tile on the 16-bit matrix
    // instruction (attention_hl_kernel); anything else: fp32 on v_mfma_f32_16x16x4_f32
    const int sp = 32 * ((S + 31) / 32);
    const int hl_lds = 4 * sp * 128 + (attn_lone_tile(S, 0, HL_WAVES) ? HL_WAVES * ATTN_PART * 4 : 0);
    if (dtype == EC_F16 && !q_prescaled && hl_lds <= 160 * 1024 && !split_force_f32()) {
        AttnHlArgs h;
        h.qkv_hi = static_cast<const _Float16 *>(qkv_hi), h.qkv_lo = static_cast<const _Float16 *>(qkv_lo);
        h.out_hi = static_cast<_Float16 *>(out_hi), h.out_lo = static_cast<_Float16 *>(out_lo);
        h.S = S, h.W = width, h.heads = heads;
        if (int rc = ec::ensure_dynamic_lds(reinterpret_cast<const void *>(attention_hl_kernel), 160 * 1024)) return rc;
        hipLaunchKernelGGL(attention_hl_kernel, dim3((unsigned)heads * (unsigned)n_seq), dim3(HL_WAVES * 64), hl_lds, s, h);
        EC_CHECK_HIP(hipGetLastError());
        return EC_OK;
    }
    // ... or whose planes fit in two passes over the keys (S <= 608): attention_hl2_kernel
    {
        const int split = (S / 2) & ~31, sl = S - split;
        const int spl = (sl & 31) == 1 ? ((sl + 15 + 15) / 16) * 16 : ((sl + 31) / 32) * 32;     // the odd key's 16 copies, else whole steps
        if (dtype == EC_F16 && !q_prescaled && split >= 32 && 4 * spl * 128 <= 160 * 1024 && (S + 15) / 16 <= HL_WAVES * HL2_TILES &&
            !split_force_f32()) {
            AttnHlArgs h;
            h.qkv_hi = static_cast<const _Float16 *>(qkv_hi), h.qkv_lo = static_cast<const _Float16 *>(qkv_lo);
            h.out_hi = static_cast<_Float16 *>(out_hi), h.out_lo = static_cast<_Float16 *>(out_lo);
            h.S = S, h.W = width, h.heads = heads;
            if (int rc = ec::ensure_dynamic_lds(reinterpret_cast<const void *>(attention_hl2_kernel), 160 * 1024)) return rc;
            hipLaunchKernelGGL(attention_hl2_kernel, dim3((unsigned)heads * (unsigned)n_seq), dim3(HL_WAVES * 64), 4 * spl * 128, s, h, split, spl);
            EC_CHECK_HIP(hipGetLastError());
            return EC_OK;
        }
    }
    const long blocks = (long)n_seq * heads * ((S + 63) / 64);
    EC_REQUIRE(blocks < (1L << 31), "ec_attention_split: %ld workgroups", blocks);
    void (*kern)(const void *, const void *, void *, void *, int, int, int, int) = nullptr;
    if (dtype == EC_F16) kern = q_prescaled ? attention_f32m_kernel<EC_F16, true, true> : attention_f32m_kernel<EC_F16, true, false>;
    else if (dtype == EC_BF16) kern = q_prescaled ? attention_f32m_kernel<EC_BF16, true, true> : attention_f32m_kernel<EC_BF16, true, false>;
    if (kern)
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, s, qkv_hi, qkv_lo, out_hi, out_lo, S, width, heads, 0);
    else
        return ec::fail(EC_ERR_INVALID, "ec_attention_split: unknown dtype %d", dtype);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

#ifdef EC_ATTN_DIAG
// (diagnostic build only; not part of the public header)
// ---- key-half workgroup pair (see attention_pair_kernel) ----
static inline int pair_split(int S) { return (S / 2) & ~31; }
static inline int pair_rows(int S)
{
    const int split = pair_split(S), sl = S - split;          // the second half is the longer one
    return (sl & 31) == 1 ? ((sl + 15 + 15) / 16) * 16 : ((sl + 31) / 32) * 32;   // its odd key's 16 copies, else whole steps
}
extern "C" __attribute__((visibility("default"))) size_t ec_attention_pair_workspace_bytes(int n_seq, int S, int heads)
{
    if (n_seq <= 0 || S <= 0 || heads <= 0) return 0;
    const size_t tiles = (size_t)n_seq * heads * ((S + 15) / 16);
    return tiles * 2 * 16 * PAIR_REC * 4 + ((tiles * 4 + 255) & ~(size_t)255);
}
extern "C" __attribute__((visibility("default"))) int ec_attention_pair(const void *qkv, void *out, int n_seq, int S, int width, int heads, int q_rows,
                                        int q_scaled, int dtype, void *workspace, size_t workspace_bytes, ec_stream_t stream)
{
    EC_REQUIRE(n_seq >= 0 && S >= 64 && heads > 0 && width == heads * 64, "ec_attention_pair: bad shape (S >= 64, head dim 64)");
    EC_REQUIRE(q_rows >= 1 && q_rows <= S, "ec_attention_pair: q_rows=%d outside 1..%d", q_rows, S);
    if (n_seq == 0) return EC_OK;
    EC_REQUIRE(qkv && out && workspace && ((uintptr_t)workspace & 255) == 0, "ec_attention_pair: null or misaligned buffer");
    const size_t need = ec_attention_pair_workspace_bytes(n_seq, S, heads);
    if (workspace_bytes < need)
        return ec::fail(EC_ERR_WORKSPACE, "ec_attention_pair: workspace %zu < %zu bytes", workspace_bytes, need);
    const int SPL = pair_rows(S), lds = SPL * 128 * 2;
    EC_REQUIRE(SPL <= 5 * 64 && lds <= 80 * 1024, "ec_attention_pair: sequence length %d needs %d bytes of LDS per workgroup (<= 81920)", S, lds);
    AttnArgs a;
    a.qkv = qkv, a.out = out, a.S = S, a.W = width, a.heads = heads, a.causal = 0, a.q_rows = q_rows, a.lse = nullptr;
    a.scale_log2e = 0.125f * 1.4426950408889634f, a.q_scaled = q_scaled;
    if (const char *dbg = getenv("EC_PAIR_DEBUG")) a.causal = atoi(dbg);      // 1: sc1 stores, 2: no exchange (timing experiments)
    const size_t tiles = (size_t)n_seq * heads * ((S + 15) / 16);
    float *part = static_cast<float *>(workspace);
    int *ticket = reinterpret_cast<int *>(static_cast<unsigned char *>(workspace) + tiles * 2 * 16 * PAIR_REC * 4);
    hipStream_t s = static_cast<hipStream_t>(stream);
    EC_CHECK_HIP(hipMemsetAsync(ticket, 0, tiles * 4, s));
    void (*kern)(const AttnArgs, float *, int *, int, int, int) = nullptr;
    if (dtype == EC_F16) kern = q_scaled ? attention_pair_kernel<EC_F16, QM_INPUT> : attention_pair_kernel<EC_F16, QM_KERNEL>;
    else if (dtype == EC_BF16) kern = q_scaled ? attention_pair_kernel<EC_BF16, QM_INPUT> : attention_pair_kernel<EC_BF16, QM_RAW>;
    else return ec::fail(EC_ERR_INVALID, "ec_attention_pair: unknown dtype %d", dtype);
    if (int rc = ec::ensure_dynamic_lds(reinterpret_cast<const void *>(kern), 80 * 1024)) return rc;
    const int n_units = n_seq * heads;
    ec::ProfScope prof(ec::PROF_ATTENTION, s, 4.0 * q_rows * S * 64.0 * heads * n_seq, (double)n_seq * width * 2.0 * (2.0 * S + 2.0 * q_rows));
    hipLaunchKernelGGL(kern, dim3((unsigned)((n_units + 7) / 8 * 16)), dim3(512), lds, s, a, part, ticket, pair_split(S), SPL, n_units);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}
#endif
