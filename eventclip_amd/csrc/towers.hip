// Host-side drivers of the two CLIP towers: enqueue the per-block kernel chain on
// the caller's stream.  No allocation, no synchronisation; scratch comes from the
// caller.  Replaces clip_model.encode_image / encode_text as the reference calls
// them (models/clip_cls.py:101, :84; structure of un-vendored openai/CLIP
// clip/model.py: ln_pre -> L x {x += MHA(ln_1 x); x += c_proj(QuickGELU(c_fc(ln_2 x)))}
// -> ln_post(CLS) @ proj, and the causal text twin ending in ln_final(EOT) @
// text_projection).
//
// Residual stream fp32; every GEMM operand 16-bit; seven launches per block:
//   LN -> GEMM(qkv) -> attention -> GEMM(out, +=x) -> LN -> GEMM(fc1, QuickGELU) -> GEMM(fc2, +=x)
#include "common.h"
#include "tower_ops.h"

namespace {

using namespace ec_tower;

struct BlockBufs {
    float *x;     // [rows, W] fp32 residual stream (folded LayerNorm: the hi plane [rows, W] 16-bit, then the lo plane)
    void *h;      // [rows, W] 16-bit: LN output / attention output
    void *qkv;    // [rows, 3W] 16-bit
    void *mlp;    // [rows, 4W] 16-bit
    float *stats; // [rows, 2] fp32: (rstd, -rstd mean) of the hi plane's rows (folded LayerNorm)
    float *sums;  // [rows, W / 64, 2] fp32: per-group (sum, sum of squares) out of the residual GEMMs' epilogues
    void *mlp_lo; // [rows, 4W] 16-bit: the lo part of the MLP activation (split-operand blocks with fp32 attention), or NULL
};

// first_only: the caller reads nothing but row 0 of every sequence after the last block (the vision
// tower: ln_post(x[:, 0]) @ proj).  That block still needs every token's keys and values, but its
// query projection, attention, out_proj, ln_2 and MLP only for row 0: n_seq rows instead of n_seq * S, through the
// same kernels with the residual stream addressed at row stride S * W.  Each output row of these
// kernels depends on its own input row only, so the class-token features are bit-identical.
int run_blocks(const ec_block_weights *blocks, int layers, int n_seq, int S, int W, int heads,
               int causal, int dtype, const BlockBufs &b, ec_stream_t s, bool first_only = false,
               bool q_scaled = false)
{
    const int rows = n_seq * S;
    for (int l = 0; l < layers; l++) {
        const ec_block_weights &w = blocks[l];
        if (first_only && l == layers - 1) {
            const long ldx = (long)S * W;
            EC_TRY(ec_layernorm(b.x, W, nullptr, w.ln1_g, w.ln1_b, rows, W, LN_EPS, b.h, W, dtype, s));
            // keys and values of every token (the k | v rows of in_proj: N = 2W into columns W .. 3W of
            // the qkv buffer), the query of the class-token rows only (N = W, one row per sequence)
            const size_t esz = 2;
            const unsigned char *wqkv = static_cast<const unsigned char *>(w.qkv_w);
            EC_TRY(gemm(rows, 2 * W, W, dtype, EC_EPI_STORE16, b.h, wqkv + (size_t)W * W * esz, w.qkv_b + W,
                        static_cast<unsigned char *>(b.qkv) + (size_t)W * esz, s, 3L * W));
            EC_TRY(gemm(n_seq, W, W, dtype, EC_EPI_STORE16, b.h, w.qkv_w, w.qkv_b, b.qkv, s, 3L * W * S,
                        (long)S * W));
            EC_TRY(q_scaled ? ec_attention_scaled_q(b.qkv, b.h, n_seq, S, W, heads, causal, 1, dtype, s)
                            : ec_attention_rows(b.qkv, b.h, n_seq, S, W, heads, causal, 1, dtype, s));
            EC_TRY(gemm(n_seq, W, W, dtype, EC_EPI_RESID32, b.h, w.out_w, w.out_b, b.x, s, ldx));
            EC_TRY(ec_layernorm(b.x, ldx, nullptr, w.ln2_g, w.ln2_b, n_seq, W, LN_EPS, b.h, W, dtype, s));
            EC_TRY(gemm(n_seq, 4 * W, W, dtype, EC_EPI_GELU16, b.h, w.fc1_w, w.fc1_b, b.mlp, s));
            EC_TRY(gemm(n_seq, W, 4 * W, dtype, EC_EPI_RESID32, b.mlp, w.fc2_w, w.fc2_b, b.x, s, ldx));
            break;
        }
        EC_TRY(ec_layernorm(b.x, W, nullptr, w.ln1_g, w.ln1_b, rows, W, LN_EPS, b.h, W, dtype, s));
        EC_TRY(gemm(rows, 3 * W, W, dtype, EC_EPI_STORE16, b.h, w.qkv_w, w.qkv_b, b.qkv, s));
        EC_TRY(q_scaled ? ec_attention_scaled_q(b.qkv, b.h, n_seq, S, W, heads, causal, S, dtype, s)
                        : ec_attention(b.qkv, b.h, n_seq, S, W, heads, causal, dtype, s));
        EC_TRY(gemm(rows, W, W, dtype, EC_EPI_RESID32, b.h, w.out_w, w.out_b, b.x, s));
        EC_TRY(ec_layernorm(b.x, W, nullptr, w.ln2_g, w.ln2_b, rows, W, LN_EPS, b.h, W, dtype, s));
        EC_TRY(gemm(rows, 4 * W, W, dtype, EC_EPI_GELU16, b.h, w.fc1_w, w.fc1_b, b.mlp, s));
        EC_TRY(gemm(rows, W, 4 * W, dtype, EC_EPI_RESID32, b.mlp, w.fc2_w, w.fc2_b, b.x, s));
    }
    return EC_OK;
}

// The same blocks with LayerNorm folded into the GEMMs around it (ec_vit_weights.ln_folded): the residual
// stream lives as hi + lo 16-bit planes, six launches per block,
//   GEMM(qkv on raw hi rows, LN in the epilogue) -> attention -> GEMM(out, (hi, lo) +=, row sums) -> merge
//   -> GEMM(fc1 on raw hi rows, LN + QuickGELU in the epilogue) -> GEMM(fc2, (hi, lo) +=, row sums) -> merge
// and the LayerNorm passes (4 + 2 bytes per element each) are gone: 16.24 -> 15.48 ms per block at the bench shape
// (tools/bench_fold.py; 15.87 with a statistics pass over the hi plane instead of the epilogue's sums), same
// rounding points.
int run_blocks_folded(const ec_block_weights *blocks, int layers, int n_seq, int S, int W, int heads, int dtype,
                      const BlockBufs &b, ec_stream_t s, bool first_only, bool q_scaled, int nsplit = 0, bool exact16 = false,
                      int nattn = 0, bool lo_fp8 = false)
{
    const int rows = n_seq * S;
    void *x_hi = b.x;
    void *x_lo = reinterpret_cast<unsigned char *>(b.x) + (size_t)rows * W * 2;
    const size_t esz = 2;
    // The statistics of a LayerNorm's input come out of the epilogue of the residual GEMM that wrote it (per-group
    // sums of the new hi values, merged by a 40 us kernel); only the first block's ln_1 reads the plane itself
    // (the embedding kernel wrote it).  Widths that are not a multiple of 64 keep the pass over the plane.
    const bool fused = W % 64 == 0;
    float *sums = fused ? b.sums : nullptr;
    const int groups = W / 64;
    // Split-operand blocks (ec_vit_weights.precise_blocks, round 5): the first nsplit blocks run on the SAME planes with
    // LayerNorm of both planes into hi + lo parts (ec_layernorm_hl), QKV and c_fc multiplying both parts (A_lo) and every
    // GEMM adding the product with its weight's lo part where it has one (W_lo): two or three MFMA products into the same
    // accumulators of ONE launch.  Plain matrices (no LayerNorm gain, no softmax scale folded in: a checkpoint stored in
    // 16 bit then has no lo parts).  In the first nattn of them attention runs in fp32 on hi + lo q, k, v.
    EC_REQUIRE(nsplit >= 0 && nsplit < layers + (first_only ? 0 : 1), "folded chain: %d split-operand blocks of %d", nsplit, layers);
    EC_REQUIRE(nattn >= 0 && nattn <= nsplit, "folded chain: %d fp32-attention blocks of %d split-operand blocks", nattn, nsplit);
    for (int l = 0; l < nsplit; l++) {
        EC_REQUIRE(blocks[l].qkv_w && blocks[l].fc1_w && blocks[l].ln1_g && blocks[l].ln2_g, "folded chain: split-operand block %d lacks its plain weights", l);
        EC_REQUIRE(exact16 || (blocks[l].qkv_w_lo && blocks[l].out_w_lo && blocks[l].fc1_w_lo && blocks[l].fc2_w_lo),
                   "folded chain: split-operand block %d has no lo weight parts (and weights_exact16 is not set)", l);
        if (lo_fp8) {
            EC_REQUIRE(blocks[l].qkv_w8 && blocks[l].fc1_w8 && blocks[l].fc2_w8, "folded chain: lo_fp8 needs the e4m3 weights (*_w8) of split-operand block %d", l);
            EC_REQUIRE((!blocks[l].qkv_w_lo || blocks[l].qkv_wlo8) && (!blocks[l].fc1_w_lo || blocks[l].fc1_wlo8),
                       "folded chain: lo_fp8 needs the e4m3 lo parts (qkv_wlo8 / fc1_wlo8) of split-operand block %d", l);
            EC_REQUIRE(b.mlp_lo && W % 128 == 0, "folded chain: lo_fp8 needs the mlp_lo buffer and a width that is a multiple of 128");
        }
    }
    if (nsplit == 0) EC_TRY(ec_row_stats(x_hi, W, rows, W, LN_EPS, b.stats, dtype, s));
    for (int l = 0; l < layers; l++) {
        const ec_block_weights &w = blocks[l];
        if (first_only && l == layers - 1) {
            // the class-token-only last block (see run_blocks): keys and values of every token, the rest for row 0
            // of every sequence, the planes addressed at row stride S * W and the statistics at stride S
            const long ldx = (long)S * W;
            const unsigned char *wqkv = static_cast<const unsigned char *>(w.qkv_w_ln);
            EC_TRY(gemm_ln(rows, 2 * W, W, dtype, EC_EPI_STORE16_LN, x_hi, wqkv + (size_t)W * W * esz, w.qkv_bf + W,
                           b.stats, 1, w.qkv_cs + W, static_cast<unsigned char *>(b.qkv) + (size_t)W * esz, s, 3L * W));
            EC_TRY(gemm_ln(n_seq, W, W, dtype, EC_EPI_STORE16_LN, x_hi, w.qkv_w_ln, w.qkv_bf, b.stats, S, w.qkv_cs,
                           b.qkv, s, 3L * W * S, ldx));
            EC_TRY(q_scaled ? ec_attention_scaled_q(b.qkv, b.h, n_seq, S, W, heads, 0, 1, dtype, s)
                            : ec_attention_rows(b.qkv, b.h, n_seq, S, W, heads, 0, 1, dtype, s));
            EC_TRY(gemm_hl(n_seq, W, W, dtype, b.h, w.out_w, w.out_b, x_hi, x_lo, s, ldx, sums));
            if (fused)
                EC_TRY(ec_row_stats_merge(sums, n_seq, groups, W, LN_EPS, b.stats, s));
            else
                EC_TRY(ec_row_stats(x_hi, ldx, n_seq, W, LN_EPS, b.stats, dtype, s));
            EC_TRY(gemm_ln(n_seq, 4 * W, W, dtype, EC_EPI_GELU16_LN, x_hi, w.fc1_w_ln, w.fc1_bf, b.stats, 1, w.fc1_cs,
                           b.mlp, s, 0, ldx));
            EC_TRY(gemm_hl(n_seq, W, 4 * W, dtype, b.mlp, w.fc2_w, w.fc2_b, x_hi, x_lo, s, ldx));
            break;
        }
        if (l < nsplit) {
            // ---- a split-operand block ----
            // lo parts live in buffers that are dead at the time: LN(x)'s in the tail of the MLP buffer (ln_1) / in the
            // qkv buffer (ln_2); q | k | v's in the head of the MLP buffer, the attention output's in its tail
            unsigned char *mlp8 = static_cast<unsigned char *>(b.mlp);
            void *h_lo1 = mlp8 + (size_t)rows * 3 * W * esz, *qkv_lo = mlp8, *att_lo = h_lo1, *h_lo2 = b.qkv;
            const bool pa = l < nattn, next_default = l + 1 >= nsplit && l + 1 < layers;
            if (lo_fp8) {
                // ---- the same block with its lo products on the FP8 matrix path (ec_vit_weights.lo_fp8) ----
                // e4m3 parts at the 16-bit byte pitch in the same dead buffers; the e4m3 copy of LN(x)'s hi part (needed where the
                // weight has a lo part) in the mlp_lo buffer (ln_1: dead until c_fc writes it) / behind the lo part in the qkv buffer (ln_2)
                unsigned char *qkv8 = static_cast<unsigned char *>(b.qkv);
                void *h8_1 = w.qkv_wlo8 ? b.mlp_lo : nullptr, *h8_2 = w.fc1_wlo8 ? qkv8 + (size_t)rows * W * esz : nullptr;
                EC_TRY(ec_layernorm_hl8(x_hi, x_lo, W, w.ln1_g, w.ln1_b, rows, W, LN_EPS, b.h, h_lo1, h8_1, W, LO8_EXP, HI8_EXP, s));
                const Fp8Parts fq = {h_lo1, w.qkv_w8, h8_1, w.qkv_wlo8, w.qkv_w8_exp, w.qkv_wlo8_exp};
                EC_TRY(gemm_split16_f8(rows, 3 * W, W, EC_EPI_STORE16, b.h, w.qkv_w, fq, w.qkv_b, b.qkv, pa ? qkv_lo : nullptr, false, s));
                if (pa) {
                    EC_TRY(ec_attention_split(b.qkv, qkv_lo, b.h, att_lo, n_seq, S, W, heads, 0, dtype, s));
                } else {
                    EC_TRY(attention_exact_scale(b.qkv, b.h, n_seq, S, W, heads, dtype, s));
                }
                EC_TRY(gemm_hl(rows, W, W, dtype, b.h, w.out_w, w.out_b, x_hi, x_lo, s, 0, nullptr, w.out_w_lo, pa ? att_lo : nullptr));
                EC_TRY(ec_layernorm_hl8(x_hi, x_lo, W, w.ln2_g, w.ln2_b, rows, W, LN_EPS, b.h, h_lo2, h8_2, W, LO8_EXP, HI8_EXP, s));
                const Fp8Parts f1 = {h_lo2, w.fc1_w8, h8_2, w.fc1_wlo8, w.fc1_w8_exp, w.fc1_wlo8_exp};
                void *m_lo8 = pa ? b.mlp_lo : nullptr;
                EC_TRY(gemm_split16_f8(rows, 4 * W, W, EC_EPI_GELU16, b.h, w.fc1_w, f1, w.fc1_b, b.mlp, m_lo8, true, s));
                if (pa) {
                    // c_proj: the activation's e4m3 lo part with the e4m3 copy of fc2_w; the weight's lo part (no e4m3 copy of the
                    // activation's hi part exists) as a 16-bit product
                    const Fp8Parts f2 = {m_lo8, w.fc2_w8, nullptr, nullptr, w.fc2_w8_exp, 0};
                    EC_TRY(gemm_hl_f8(rows, W, 4 * W, b.mlp, w.fc2_w, f2, w.fc2_w_lo, w.fc2_b, x_hi, x_lo, s, next_default ? sums : nullptr));
                } else {
                    EC_TRY(gemm_hl(rows, W, 4 * W, dtype, b.mlp, w.fc2_w, w.fc2_b, x_hi, x_lo, s, 0, next_default ? sums : nullptr, w.fc2_w_lo,
                                   nullptr));
                }
            } else {
            EC_TRY(ec_layernorm_hl(x_hi, x_lo, W, w.ln1_g, w.ln1_b, rows, W, LN_EPS, b.h, h_lo1, W, dtype, s));
            EC_TRY(gemm_split16(rows, 3 * W, W, dtype, EC_EPI_STORE16, b.h, h_lo1, w.qkv_w, w.qkv_w_lo, w.qkv_b, b.qkv,
                                pa ? qkv_lo : nullptr, s));
            if (pa) {
                EC_TRY(ec_attention_split(b.qkv, qkv_lo, b.h, att_lo, n_seq, S, W, heads, 0, dtype, s));
            } else {
                EC_TRY(attention_exact_scale(b.qkv, b.h, n_seq, S, W, heads, dtype, s));
            }
            EC_TRY(gemm_hl(rows, W, W, dtype, b.h, w.out_w, w.out_b, x_hi, x_lo, s, 0, nullptr, w.out_w_lo, pa ? att_lo : nullptr));
            EC_TRY(ec_layernorm_hl(x_hi, x_lo, W, w.ln2_g, w.ln2_b, rows, W, LN_EPS, b.h, h_lo2, W, dtype, s));
            // (with fp32 attention also the MLP activation as hi + lo parts into c_proj: where attention is sharp the 16-bit
            // rounding of QuickGELU's output in the first blocks is the next contribution behind q / k)
            void *m_lo = pa ? b.mlp_lo : nullptr;
            EC_TRY(gemm_split16(rows, 4 * W, W, dtype, EC_EPI_GELU16, b.h, h_lo2, w.fc1_w, w.fc1_w_lo, w.fc1_b, b.mlp, m_lo, s));
            // the first default block behind the split-operand blocks takes its statistics from this epilogue's sums
            EC_TRY(gemm_hl(rows, W, 4 * W, dtype, b.mlp, w.fc2_w, w.fc2_b, x_hi, x_lo, s, 0, next_default ? sums : nullptr,
                           w.fc2_w_lo, m_lo));
            }
            if (next_default) {
                if (fused)
                    EC_TRY(ec_row_stats_merge(sums, rows, groups, W, LN_EPS, b.stats, s));
                else
                    EC_TRY(ec_row_stats(x_hi, W, rows, W, LN_EPS, b.stats, dtype, s));
            }
            continue;
        }
        EC_TRY(gemm_ln(rows, 3 * W, W, dtype, EC_EPI_STORE16_LN, x_hi, w.qkv_w_ln, w.qkv_bf, b.stats, 1, w.qkv_cs, b.qkv, s));
        EC_TRY(q_scaled ? ec_attention_scaled_q(b.qkv, b.h, n_seq, S, W, heads, 0, S, dtype, s)
                        : ec_attention(b.qkv, b.h, n_seq, S, W, heads, 0, dtype, s));
        EC_TRY(gemm_hl(rows, W, W, dtype, b.h, w.out_w, w.out_b, x_hi, x_lo, s, 0, sums));
        if (fused)
            EC_TRY(ec_row_stats_merge(sums, rows, groups, W, LN_EPS, b.stats, s));
        else
            EC_TRY(ec_row_stats(x_hi, W, rows, W, LN_EPS, b.stats, dtype, s));
        EC_TRY(gemm_ln(rows, 4 * W, W, dtype, EC_EPI_GELU16_LN, x_hi, w.fc1_w_ln, w.fc1_bf, b.stats, 1, w.fc1_cs, b.mlp, s));
        EC_TRY(gemm_hl(rows, W, 4 * W, dtype, b.mlp, w.fc2_w, w.fc2_b, x_hi, x_lo, s, 0, l + 1 < layers ? sums : nullptr));
        if (l + 1 < layers) {
            if (fused)
                EC_TRY(ec_row_stats_merge(sums, rows, groups, W, LN_EPS, b.stats, s));
            else
                EC_TRY(ec_row_stats(x_hi, W, rows, W, LN_EPS, b.stats, dtype, s));
        }
    }
    return EC_OK;
}

// Split-precision chain: every GEMM is xh.wh + xh.wl + xl.wh accumulated in fp32, attention and
// QuickGELU run in fp32, and every activation that feeds a GEMM is carried as hi + lo parts.
struct PreciseBufs {
    float *x;           // [rows, W] fp32 residual stream
    void *h_hi, *h_lo;  // [rows, W] LN / attention output, split
    float *wide;        // [rows, 4W] fp32: qkv (3W) or the c_fc output (4W)
    void *m_hi, *m_lo;  // [rows, 4W] QuickGELU output, split
};

int run_blocks_precise(const ec_block_weights *blocks, int layers, int n_seq, int S, int W, int heads,
                       int causal, int dtype, const PreciseBufs &b, ec_stream_t s, bool exact16 = false)
{
    const int rows = n_seq * S;
    for (int l = 0; l < layers; l++) {
        const ec_block_weights &w = blocks[l];
        // exact16 (ec_vit_weights.weights_exact16): a NULL lo part says that the matrix is its 16-bit value
        EC_REQUIRE(exact16 || (w.qkv_w_lo && w.out_w_lo && w.fc1_w_lo && w.fc2_w_lo),
                   "precise tower: block %d has no lo weight parts", l);
        EC_TRY(ec_layernorm_split(b.x, W, nullptr, w.ln1_g, w.ln1_b, rows, W, LN_EPS, b.h_hi, b.h_lo,
                                  W, dtype, s));
        if (!causal) {
            // the image tower: q | k | v leave the GEMM as hi + lo 16-bit parts (12 of the wide buffer's 16 bytes per row
            // element) and attention runs on them -- on the 16-bit matrix instruction where the four planes fit the LDS
            void *qkv_hi = b.wide, *qkv_lo = reinterpret_cast<unsigned char *>(b.wide) + (size_t)rows * 3 * W * 2;
            EC_TRY(gemm_split16(rows, 3 * W, W, dtype, EC_EPI_STORE16, b.h_hi, b.h_lo, w.qkv_w, w.qkv_w_lo, w.qkv_b, qkv_hi, qkv_lo, s));
            EC_TRY(ec_attention_split(qkv_hi, qkv_lo, b.h_hi, b.h_lo, n_seq, S, W, heads, 0, dtype, s));
        } else {
            EC_TRY(gemm3(rows, 3 * W, W, dtype, false, b.h_hi, b.h_lo, w.qkv_w, w.qkv_w_lo, w.qkv_b,
                         b.wide, s));
            EC_TRY(ec_attention_f32(b.wide, b.h_hi, b.h_lo, n_seq, S, W, heads, causal, dtype, s));
        }
        EC_TRY(gemm3(rows, W, W, dtype, true, b.h_hi, b.h_lo, w.out_w, w.out_w_lo, w.out_b, b.x, s));
        EC_TRY(ec_layernorm_split(b.x, W, nullptr, w.ln2_g, w.ln2_b, rows, W, LN_EPS, b.h_hi, b.h_lo,
                                  W, dtype, s));
        EC_TRY(gemm3(rows, 4 * W, W, dtype, false, b.h_hi, b.h_lo, w.fc1_w, w.fc1_w_lo, w.fc1_b,
                     b.wide, s));
        EC_TRY(ec_split16(b.wide, (long)rows * 4 * W, 1, b.m_hi, b.m_lo, dtype, s));
        EC_TRY(gemm3(rows, W, 4 * W, dtype, true, b.m_hi, b.m_lo, w.fc2_w, w.fc2_w_lo, w.fc2_b, b.x,
                     s));
    }
    return EC_OK;
}

size_t carve_precise(Scratch &sc, int chunk, int S, int W, PreciseBufs &b, void **s_hi, void **s_lo,
                     int **idx)
{
    const size_t rows = (size_t)chunk * S;
    b.x = (float *)sc.take(rows * W * 4);
    b.h_hi = sc.take(rows * W * 2);
    b.h_lo = sc.take(rows * W * 2);
    b.wide = (float *)sc.take(rows * 4 * W * 4);
    b.m_hi = sc.take(rows * 4 * W * 2);
    b.m_lo = sc.take(rows * 4 * W * 2);
    *s_hi = sc.take((size_t)chunk * W * 2);
    *s_lo = sc.take((size_t)chunk * W * 2);
    *idx = (int *)sc.take((size_t)chunk * 4);
    return sc.off;
}

// carve the scratch for `chunk` sequences of length S; patch_rows > 0 adds the fp32
// patch-GEMM output (aliased onto the mlp buffer: both are dead at the same time)
size_t carve(Scratch &sc, int chunk, int S, int W, int out_rows_extra, BlockBufs &b, void **small16,
             int **idx, void **small16_lo = nullptr, bool with_mlp_lo = false)
{
    const size_t rows = (size_t)chunk * S;
    b.x = (float *)sc.take(rows * W * 4);
    b.h = sc.take(rows * W * 2);
    b.qkv = sc.take(rows * 3 * W * 2);
    b.mlp = sc.take(rows * 4 * W * 2);   // >= rows * W * 4 bytes: also holds the patch GEMM output
    b.stats = (float *)sc.take(rows * 8 + 16);   // (+ one pair: the LN epilogues fetch the pairs two at a time)
    b.sums = (float *)sc.take(rows * (size_t)(W / 64 + 1) * 8);
    b.mlp_lo = with_mlp_lo ? sc.take(rows * 4 * W * 2) : nullptr;
    *small16 = sc.take((size_t)chunk * W * 2);
    void *lo = sc.take((size_t)chunk * W * 2);
    if (small16_lo) *small16_lo = lo;
    *idx = (int *)sc.take((size_t)chunk * 4 + (size_t)out_rows_extra);
    return sc.off;
}

__global__ void eot_index_kernel(const int *tokens, int n_txt, int ctx, int *idx)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_txt) return;
    const int *t = tokens + (long)n * ctx;
    int best = 0, bv = t[0];
    for (int s = 1; s < ctx; s++)
        if (t[s] > bv) bv = t[s], best = s;   // first maximum, as torch.argmax
    idx[n] = n * ctx + best;
}

}  // namespace

extern "C" {

EC_API size_t ec_vit_workspace_bytes(const ec_vit_weights *w, int chunk)
{
    if (!w || chunk <= 0) return 0;
    const int g = w->image_size / w->patch;
    Scratch sc{nullptr, 0, 0};
    void *s16, *s16b;
    int *idx;
    if (w->precise) {
        PreciseBufs pb;
        return carve_precise(sc, chunk, g * g + 1, w->width, pb, &s16, &s16b, &idx);
    }
    BlockBufs b;
    carve(sc, chunk, g * g + 1, w->width, 0, b, &s16, &idx, nullptr, w->precise_blocks > 0 && (w->precise_attn_blocks > 0 || w->lo_fp8));
    return sc.off + (w->low_latency ? LATENCY_WS_BYTES : 0);
}

EC_API size_t ec_text_workspace_bytes(const ec_text_weights *w, int chunk)
{
    if (!w || chunk <= 0) return 0;
    Scratch sc{nullptr, 0, 0};
    void *s16, *s16b;
    int *idx;
    if (w->precise) {
        PreciseBufs pb;
        return carve_precise(sc, chunk, w->ctx, w->width, pb, &s16, &s16b, &idx);
    }
    BlockBufs b;
    return carve(sc, chunk, w->ctx, w->width, 0, b, &s16, &idx);
}

EC_API int ec_vit_encode(const ec_vit_weights *w, const void *patches, int n_img, float *feats,
                         void *workspace, size_t workspace_bytes, int chunk, ec_stream_t stream)
{
    EC_REQUIRE(w && w->blocks, "ec_vit_encode: weights are null");
    EC_REQUIRE(n_img >= 0 && chunk > 0, "ec_vit_encode: n_img=%d chunk=%d", n_img, chunk);
    if (n_img == 0) return EC_OK;
    EC_REQUIRE(patches && feats && workspace, "ec_vit_encode: null buffer");
    EC_REQUIRE(w->image_size % w->patch == 0, "ec_vit_encode: image %d not a multiple of patch %d",
               w->image_size, w->patch);
    EC_REQUIRE(w->width == w->heads * 64, "ec_vit_encode: head dim must be 64");
    EC_REQUIRE(w->kpad % 64 == 0 && w->kpad >= 6 * w->patch * w->patch, "ec_vit_encode: bad kpad %d",
               w->kpad);
    EC_REQUIRE(w->conv_w && w->proj_w && (w->weights_exact16 || (w->conv_w_lo && w->proj_w_lo)),
               "ec_vit_encode: conv / proj weights need their hi and lo parts (a lo part may be NULL with weights_exact16)");
    EC_REQUIRE(w->out_dim % 16 == 0, "ec_vit_encode: out_dim %d", w->out_dim);
    EC_REQUIRE(!(w->precise && w->q_scaled), "ec_vit_encode: the split-precision tower takes a plain q (q_scaled = 0)");
    const int g = w->image_size / w->patch, G = g * g, S = G + 1, W = w->width, dt = w->dtype;
    if (chunk > n_img) chunk = n_img;
    Scratch sc{(unsigned char *)workspace, 0, workspace_bytes};
    const size_t esz = 2;
    if (w->precise) {
        PreciseBufs pb;
        void *c_hi, *c_lo;
        int *pidx;
        const size_t pneed = carve_precise(sc, chunk, S, W, pb, &c_hi, &c_lo, &pidx);
        if (pneed > workspace_bytes)
            return ec::fail(EC_ERR_WORKSPACE, "ec_vit_encode: workspace %zu < %zu bytes",
                            workspace_bytes, pneed);
        for (int i0 = 0; i0 < n_img; i0 += chunk) {
            const int n = (n_img - i0 < chunk) ? n_img - i0 : chunk;
            const unsigned char *p = (const unsigned char *)patches + (size_t)i0 * G * w->kpad * esz;
            EC_TRY(patch_embed(w, p, n * G, pb.wide, stream));
            EC_TRY(ec_vit_embed(pb.wide, w->cls, w->pos, w->ln_pre_g, w->ln_pre_b, n, S, W, LN_EPS,
                                pb.x, stream));
            EC_TRY(run_blocks_precise(w->blocks, w->layers, n, S, W, w->heads, 0, dt, pb, stream, w->weights_exact16 != 0));
            EC_TRY(ec_layernorm_split(pb.x, (long)S * W, nullptr, w->ln_post_g, w->ln_post_b, n, W,
                                      LN_EPS, c_hi, c_lo, W, dt, stream));
            EC_TRY(gemm3(n, w->out_dim, W, dt, false, c_hi, c_lo, w->proj_w, w->proj_w_lo, nullptr,
                         feats + (size_t)i0 * w->out_dim, stream));
        }
        return EC_OK;
    }
    BlockBufs b;
    void *cls16, *cls16_lo;
    int *idx;
    size_t need = carve(sc, chunk, S, W, 0, b, &cls16, &idx, &cls16_lo, w->precise_blocks > 0 && (w->precise_attn_blocks > 0 || w->lo_fp8));
    // precise_blocks: the first blocks of the folded chain multiply both planes of the residual stream and the
    // weights' lo parts (run_blocks_folded)
    const int pblocks = w->precise_blocks;
    if (pblocks != 0) {
        EC_REQUIRE(pblocks > 0 && pblocks < w->layers && w->ln_folded && !w->low_latency,
                   "ec_vit_encode: precise_blocks=%d needs 0 < precise_blocks < layers=%d, ln_folded and no low_latency",
                   pblocks, w->layers);
        // the lo plane of the stream is fp16 whatever the operand type: only an f16 tower can multiply it
        EC_REQUIRE(dt == EC_F16, "ec_vit_encode: precise_blocks needs dtype EC_F16 (the lo plane of the residual stream is fp16)");
    }
    // low-latency mode: under-filled GEMM launches (a few frames) run K-batched through this scratch
    struct ScratchGuard {
        ~ScratchGuard() { latency_scratch() = {nullptr, 0}; }
    } guard;
    if (w->low_latency) {
        latency_scratch() = {sc.take(LATENCY_WS_BYTES), LATENCY_WS_BYTES};
        need = sc.off;
    }
    if (need > workspace_bytes)
        return ec::fail(EC_ERR_WORKSPACE, "ec_vit_encode: workspace %zu < %zu bytes", workspace_bytes,
                        need);
    const bool folded = w->ln_folded && !w->low_latency;
    if (folded)
        for (int l = pblocks; l < w->layers; l++)
            EC_REQUIRE(w->blocks[l].qkv_w_ln && w->blocks[l].qkv_cs && w->blocks[l].qkv_bf && w->blocks[l].fc1_w_ln &&
                           w->blocks[l].fc1_cs && w->blocks[l].fc1_bf,
                       "ec_vit_encode: ln_folded but block %d lacks its folded weights", l);
    for (int i0 = 0; i0 < n_img; i0 += chunk) {
        const int n = (n_img - i0 < chunk) ? n_img - i0 : chunk;
        const unsigned char *p = (const unsigned char *)patches + (size_t)i0 * G * w->kpad * esz;
        float *patch_out = (float *)b.mlp;
        EC_TRY(patch_embed(w, p, n * G, patch_out, stream));
        if (folded) {
            // residual stream as hi + lo planes in the fp32 stream's 4 bytes per element
            void *x_hi = b.x, *x_lo = reinterpret_cast<unsigned char *>(b.x) + (size_t)n * S * W * 2;
            EC_TRY(vit_embed_hl(patch_out, w->cls, w->pos, w->ln_pre_g, w->ln_pre_b, n, S, W, LN_EPS, x_hi, x_lo, dt, stream));
            EC_TRY(run_blocks_folded(w->blocks, w->layers, n, S, W, w->heads, dt, b, stream, w->full_last_block == 0,
                                     w->q_scaled != 0, pblocks, w->weights_exact16 != 0,
                                     w->precise_attn_blocks < pblocks ? w->precise_attn_blocks : pblocks, pblocks > 0 && w->lo_fp8 != 0));
            // the class rows back to fp32 (x = hi + lo) for ln_post; patch_out (the mlp buffer) is free by now
            EC_TRY(join_hl_rows(x_hi, x_lo, (long)S * W, n, W, patch_out, dt, stream));
            EC_TRY(ec_layernorm_split(patch_out, W, nullptr, w->ln_post_g, w->ln_post_b, n, W, LN_EPS, cls16, cls16_lo, W,
                                      dt, stream));
            EC_TRY(gemm3(n, w->out_dim, W, dt, false, cls16, cls16_lo, w->proj_w, w->proj_w_lo, nullptr,
                         feats + (size_t)i0 * w->out_dim, stream));
            continue;
        }
        EC_TRY(ec_vit_embed(patch_out, w->cls, w->pos, w->ln_pre_g, w->ln_pre_b, n, S, W, LN_EPS, b.x,
                            stream));
        EC_TRY(run_blocks(w->blocks, w->layers, n, S, W, w->heads, 0, dt, b, stream,
                          w->full_last_block == 0, w->q_scaled != 0));
        // ln_post on the CLS rows (row stride S*W), then @ proj.  These n rows are the features
        // themselves: their 16-bit rounding is not averaged over anything downstream and was 45 % of the
        // logit error budget (tools/rounding_budget.py), so both operands keep their lo parts here
        // (three launches over n rows: free).
        EC_TRY(ec_layernorm_split(b.x, (long)S * W, nullptr, w->ln_post_g, w->ln_post_b, n, W, LN_EPS,
                                  cls16, cls16_lo, W, dt, stream));
        EC_TRY(gemm3(n, w->out_dim, W, dt, false, cls16, cls16_lo, w->proj_w, w->proj_w_lo, nullptr,
                     feats + (size_t)i0 * w->out_dim, stream));
    }
    return EC_OK;
}

EC_API int ec_text_encode(const ec_text_weights *w, const int32_t *tokens, int n_txt, float *feats,
                          void *workspace, size_t workspace_bytes, int chunk, ec_stream_t stream)
{
    EC_REQUIRE(w && w->blocks, "ec_text_encode: weights are null");
    EC_REQUIRE(n_txt >= 0 && chunk > 0, "ec_text_encode: n_txt=%d chunk=%d", n_txt, chunk);
    if (n_txt == 0) return EC_OK;
    EC_REQUIRE(tokens && feats && workspace, "ec_text_encode: null buffer");
    EC_REQUIRE(w->width == w->heads * 64, "ec_text_encode: head dim must be 64");
    EC_REQUIRE(w->out_dim % 16 == 0, "ec_text_encode: out_dim %d", w->out_dim);
    const int S = w->ctx, W = w->width, dt = w->dtype;
    if (chunk > n_txt) chunk = n_txt;
    Scratch sc{(unsigned char *)workspace, 0, workspace_bytes};
    hipStream_t hs = static_cast<hipStream_t>(stream);
    if (w->precise) {
        EC_REQUIRE(w->proj_w_lo, "ec_text_encode: precise tower without lo weights");
        PreciseBufs pb;
        void *e_hi, *e_lo;
        int *pidx;
        const size_t pneed = carve_precise(sc, chunk, S, W, pb, &e_hi, &e_lo, &pidx);
        if (pneed > workspace_bytes)
            return ec::fail(EC_ERR_WORKSPACE, "ec_text_encode: workspace %zu < %zu bytes",
                            workspace_bytes, pneed);
        for (int i0 = 0; i0 < n_txt; i0 += chunk) {
            const int n = (n_txt - i0 < chunk) ? n_txt - i0 : chunk;
            const int32_t *tok = tokens + (size_t)i0 * S;
            EC_TRY(ec_text_embed(tok, w->token_embedding, w->pos, n, S, W, w->vocab, pb.x, stream));
            EC_TRY(run_blocks_precise(w->blocks, w->layers, n, S, W, w->heads, 1, dt, pb, stream));
            hipLaunchKernelGGL(eot_index_kernel, dim3((n + 255) / 256), dim3(256), 0, hs, tok, n, S,
                               pidx);
            EC_CHECK_HIP(hipGetLastError());
            EC_TRY(ec_layernorm_split(pb.x, W, pidx, w->ln_final_g, w->ln_final_b, n, W, LN_EPS, e_hi,
                                      e_lo, W, dt, stream));
            EC_TRY(gemm3(n, w->out_dim, W, dt, false, e_hi, e_lo, w->proj_w, w->proj_w_lo, nullptr,
                         feats + (size_t)i0 * w->out_dim, stream));
        }
        return EC_OK;
    }
    BlockBufs b;
    void *eot16;
    int *idx;
    const size_t need = carve(sc, chunk, S, W, 0, b, &eot16, &idx);
    if (need > workspace_bytes)
        return ec::fail(EC_ERR_WORKSPACE, "ec_text_encode: workspace %zu < %zu bytes",
                        workspace_bytes, need);
    for (int i0 = 0; i0 < n_txt; i0 += chunk) {
        const int n = (n_txt - i0 < chunk) ? n_txt - i0 : chunk;
        const int32_t *tok = tokens + (size_t)i0 * S;
        EC_TRY(ec_text_embed(tok, w->token_embedding, w->pos, n, S, W, w->vocab, b.x, stream));
        EC_TRY(run_blocks(w->blocks, w->layers, n, S, W, w->heads, 1, dt, b, stream));
        hipLaunchKernelGGL(eot_index_kernel, dim3((n + 255) / 256), dim3(256), 0, hs, tok, n, S, idx);
        EC_CHECK_HIP(hipGetLastError());
        EC_TRY(ec_layernorm(b.x, W, idx, w->ln_final_g, w->ln_final_b, n, W, LN_EPS, eot16, W, dt,
                            stream));
        EC_TRY(gemm(n, w->out_dim, W, dt, EC_EPI_STORE32, eot16, w->proj_w, nullptr,
                    feats + (size_t)i0 * w->out_dim, stream));
    }
    return EC_OK;
}

}  // extern "C"
