// Backward of the fused multi-head self-attention (gfx950, head dim 64): what torch autograd
// computes for nn.MultiheadAttention inside CLIP's ResidualAttentionBlocks when the reference
// fine-tunes the vision tower (models/clip_cls_ft.py:44-80 marks visual.* trainable,
// models/lora.py:160-330 is the same attention with LoRA-merged projection weights).
//
// With P = softmax(Q K^T / 8) (recomputed from the saved log-sum-exp of the forward pass),
// dP = dO V^T, D_i = <dO_i, O_i>, dS = P o (dP - D):
//     dQ = dS K / 8        dK = dS^T Q / 8        dV = P^T dO
// dQ reduces over keys and dK / dV over queries, so the score tile is needed in both orientations.
// Sequences are short (50 .. 577 tokens): the tile is simply recomputed by two kernels, each the
// forward kernel's shape with one workgroup per (sequence, head) and no cross-wave reduction:
//   attention_dq_kernel   a wave owns 16 queries (lane = query column), keys come from LDS:
//                         S^T = K Q^T, dP^T = V dO^T, dQ^T += K^T dS^T (K read transposed from LDS)
//   attention_dkv_kernel  a wave owns 16 keys (lane = key column), queries come from LDS:
//                         S = Q K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS
// The exponentiated-score registers are the B operand of the second MFMA, as in the forward kernel.
// LDS rows past 288 are staged in further chunks (S = 577 at 336 px), the accumulators stay in
// registers across chunks.
#include "common.h"
#include "mfma.h"

namespace {

using namespace ec;

constexpr int CHUNK = 288;   // rows of a staged operand per pass: 4 images x 36 KiB (dK / dV kernel)

struct BwdArgs {
    const void *qkv;    // [n_seq * S, 3W] 16-bit: q | k | v
    const void *out;    // [n_seq * S, W] forward output O
    const void *dout;   // [n_seq * S, W] gradient of O
    const float *lse;   // [n_seq, heads, S] from ec_attention_train
    float *delta;       // [n_seq, heads, S] D_i = <dO_i, O_i> (written by the dQ kernel, read by dK / dV)
    void *dqkv;         // [n_seq * S, 3W] 16-bit (out)
    int S, W, heads;
    float scale_log2e;
};

// row-major image with 128-B rows, 16-B chunk c at (c ^ (row & 7)): conflict-free ds_read_b128 of
// 16 consecutive rows (the MFMA operand whose contraction index runs along the row)
__device__ __forceinline__ void put_rows(unsigned char *lds, int row, int ch, u32x4 v)
{
    *reinterpret_cast<u32x4 *>(lds + row * 128 + ((ch ^ (row & 7)) << 4)) = v;
}
// row-major image for transposed reads (ds_read_b64_tr_b16): 8-byte slot u at u ^ ((row >> 1) & 3)
__device__ __forceinline__ void put_cols(unsigned char *lds, int row, int ch, u32x4 v)
{
    if ((row >> 1) & 1) v = u32x4{v[2], v[3], v[0], v[1]};
    *reinterpret_cast<u32x4 *>(lds + row * 128 + ((ch ^ ((row >> 2) & 1)) << 4)) = v;
}

// A-operand fragment whose 16 rows are rows row0 + c16 of a put_rows image, k = 32 ks + 8 g + j
template <int DT>
__device__ __forceinline__ typename T16<DT>::v8 get_rows(const unsigned char *lds, int row, int ks, int g)
{
    return *reinterpret_cast<const typename T16<DT>::v8 *>(lds + row * 128 + (((ks * 4 + g) ^ (row & 7)) << 4));
}
// A-operand fragment of the TRANSPOSE of a put_cols image over rows r0 .. r0 + 31: MFMA row i <-> column
// (i >> 2) * 16 + 4 dt + (i & 3), k index j <-> row r0 + 16 (j >> 2) + 4 g + (j & 3) -- the order the score
// registers have as a B operand
template <int DT>
__device__ __forceinline__ typename T16<DT>::v8 get_cols(const unsigned char *lds, int r0, int dt, int g, int c16)
{
    typedef typename T16<DT>::v4 v4;
    typename T16<DT>::v8 f;
#pragma unroll
    for (int hh = 0; hh < 2; hh++) {
        const int row = r0 + 16 * hh + 4 * g + (c16 >> 2);
        const int u = ((c16 & 3) * 4 + dt) ^ ((row >> 1) & 3);
        const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4 *)(lds + row * 128 + u * 8));
        const v4 tv = __builtin_bit_cast(v4, t);
        f[4 * hh] = tv[0], f[4 * hh + 1] = tv[1], f[4 * hh + 2] = tv[2], f[4 * hh + 3] = tv[3];
    }
    return f;
}

// ------------------------------------------------------------------------------------------
// dQ.  LDS: K (rows), V (rows), K (cols) of one chunk of keys.
// ------------------------------------------------------------------------------------------
template <int DT, int KSTEPS>
__device__ __forceinline__ void dq_block(const unsigned char *ldsK, const unsigned char *ldsV,
                                         const unsigned char *ldsKt, int key0, int kbase, int S,
                                         const typename T16<DT>::v8 (&qf)[2],
                                         const typename T16<DT>::v8 (&dof)[2], float scale_log2e, float lse2,
                                         float delta, f32x4 (&dq)[4], int g, int c16)
{
    typedef typename T16<DT>::elem elem;
    typedef typename T16<DT>::v8 v8;
    constexpr int NT = 2 * KSTEPS;
    f32x4 sc[NT], dp[NT];
#pragma unroll
    for (int kt = 0; kt < NT; kt++) {
        sc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        dp[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int row = key0 + kt * 16 + c16;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            sc[kt] = mfma16(get_rows<DT>(ldsK, row, ks, g), qf[ks], sc[kt]);
            dp[kt] = mfma16(get_rows<DT>(ldsV, row, ks, g), dof[ks], dp[kt]);
        }
    }
    // dS^T[key][q] = P (dP - D), P = exp2(s * scale - lse2); keys past the sequence contribute nothing
#pragma unroll
    for (int kt = 0; kt < NT; kt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int key = kbase + key0 + kt * 16 + 4 * g + r;
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kt][r], scale_log2e, -lse2));
            sc[kt][r] = key < S ? p * (dp[kt][r] - delta) : 0.f;
        }
#pragma unroll
    for (int s = 0; s < KSTEPS; s++) {
        v8 pf;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            pf[r] = to16(sc[2 * s][r], elem());
            pf[4 + r] = to16(sc[2 * s + 1][r], elem());
        }
#pragma unroll
        for (int dt = 0; dt < 4; dt++)
            dq[dt] = mfma16(get_cols<DT>(ldsKt, key0 + 32 * s, dt, g, c16), pf, dq[dt]);
    }
}

template <int DT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void attention_dq_kernel(const BwdArgs a)
{
    typedef typename T16<DT>::elem elem;
    typedef typename T16<DT>::v8 v8;
    constexpr int THREADS = WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int S = a.S, W = a.W;
    const int SP = 32 * ((S + 31) / 32);
    const int CP = SP < CHUNK ? SP : CHUNK;                 // rows per staged chunk
    unsigned char *ldsK = smem, *ldsV = smem + CP * 128, *ldsKt = smem + 2 * CP * 128;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c16 = lane & 15;
    const int head = blockIdx.x % a.heads, seq = blockIdx.x / a.heads;
    const long ld = 3L * W;
    const elem *base = (const elem *)a.qkv + (long)seq * S * ld + head * 64;
    const elem *obase = (const elem *)a.out + (long)seq * S * W + head * 64;
    const elem *dobase = (const elem *)a.dout + (long)seq * S * W + head * 64;
    const long stat = ((long)seq * a.heads + head) * S;

    const int n_qt = (S + 15) / 16;
    const int passes = (n_qt + WAVES - 1) / WAVES;
    for (int pass = 0; pass < passes; pass++) {
        const int qt = pass * WAVES + wave;
        const bool active = qt < n_qt;
        const int qrow = qt * 16 + c16;
        const int qsrc = qrow < S ? qrow : S - 1;
        v8 qf[2], dof[2];
        float lse2 = 0.f, delta = 0.f;
        if (active) {
            float part = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                qf[ks] = *reinterpret_cast<const v8 *>(base + (long)qsrc * ld + ks * 32 + g * 8);
                dof[ks] = *reinterpret_cast<const v8 *>(dobase + (long)qsrc * W + ks * 32 + g * 8);
                const v8 of = *reinterpret_cast<const v8 *>(obase + (long)qsrc * W + ks * 32 + g * 8);
#pragma unroll
                for (int j = 0; j < 8; j++) part = __builtin_fmaf((float)dof[ks][j], (float)of[j], part);
            }
            delta = xor_sum(part);
            lse2 = a.lse[stat + qsrc];
            if (g == 0 && qrow < S) a.delta[stat + qrow] = delta;
        }
        f32x4 dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; dt++) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kbase = 0; kbase < SP; kbase += CP) {
            const int rows = SP - kbase < CP ? SP - kbase : CP;
            if (pass > 0 || kbase > 0) __syncthreads();     // every wave is done with the previous image
            if (pass == 0 || SP > CP) {
                const int r_in = threadIdx.x >> 3, ch = threadIdx.x & 7;
                for (int row = r_in; row < rows; row += THREADS / 8) {
                    const int srow = kbase + row < S ? kbase + row : S - 1;
                    const elem *src = base + (long)srow * ld + ch * 8;
                    const u32x4 kv = *reinterpret_cast<const u32x4 *>(src + W);
                    const u32x4 vv = *reinterpret_cast<const u32x4 *>(src + 2 * W);
                    put_rows(ldsK, row, ch, kv);
                    put_rows(ldsV, row, ch, vv);
                    put_cols(ldsKt, row, ch, kv);
                }
            }
            __syncthreads();
            if (active) {
                int key0 = 0;
                for (; key0 + 64 <= rows; key0 += 64)
                    dq_block<DT, 2>(ldsK, ldsV, ldsKt, key0, kbase, S, qf, dof, a.scale_log2e, lse2, delta, dq, g,
                                    c16);
                if (key0 < rows)
                    dq_block<DT, 1>(ldsK, ldsV, ldsKt, key0, kbase, S, qf, dof, a.scale_log2e, lse2, delta, dq, g,
                                    c16);
            }
        }
        // dQ = dS K / 8: lane owns query c16, head dims 16 g .. 16 g + 15
        if (active && qrow < S) {
            elem ov[16];
#pragma unroll
            for (int dt = 0; dt < 4; dt++)
#pragma unroll
                for (int r = 0; r < 4; r++) ov[4 * dt + r] = to16(dq[dt][r] * 0.125f, elem());
            elem *dst = (elem *)a.dqkv + ((long)seq * S + qrow) * ld + head * 64 + g * 16;
            *reinterpret_cast<u32x4 *>(dst) = *reinterpret_cast<const u32x4 *>(&ov[0]);
            *reinterpret_cast<u32x4 *>(dst + 8) = *reinterpret_cast<const u32x4 *>(&ov[8]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// dK, dV.  LDS: Q (rows), dO (rows), Q (cols), dO (cols) of one chunk of queries, their lse / D.
// ------------------------------------------------------------------------------------------
template <int DT, int KSTEPS>
__device__ __forceinline__ void dkv_block(const unsigned char *ldsQ, const unsigned char *ldsD,
                                          const unsigned char *ldsQt, const unsigned char *ldsDt,
                                          const float *lse_s, const float *delta_s, int q0,
                                          const typename T16<DT>::v8 (&kf)[2],
                                          const typename T16<DT>::v8 (&vf)[2], float scale_log2e,
                                          f32x4 (&dk)[4], f32x4 (&dv)[4], int g, int c16)
{
    typedef typename T16<DT>::elem elem;
    typedef typename T16<DT>::v8 v8;
    constexpr int NT = 2 * KSTEPS;
    f32x4 sc[NT], dp[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        sc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int row = q0 + t * 16 + c16;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            sc[t] = mfma16(get_rows<DT>(ldsQ, row, ks, g), kf[ks], sc[t]);
            dp[t] = mfma16(get_rows<DT>(ldsD, row, ks, g), vf[ks], dp[t]);
        }
    }
    // sc[t][r] <-> query q0 + 16 t + 4 g + r, this lane's key: P, then dS = P (dP - D_q)
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const f32x4 l4 = *reinterpret_cast<const f32x4 *>(lse_s + q0 + t * 16 + 4 * g);
        const f32x4 d4 = *reinterpret_cast<const f32x4 *>(delta_s + q0 + t * 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[t][r], scale_log2e, -l4[r]));
            sc[t][r] = p;                         // padded queries carry lse = +inf: p = 0
            dp[t][r] = p * (dp[t][r] - d4[r]);
        }
    }
#pragma unroll
    for (int s = 0; s < KSTEPS; s++) {
        v8 pf, sf;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            pf[r] = to16(sc[2 * s][r], elem());
            pf[4 + r] = to16(sc[2 * s + 1][r], elem());
            sf[r] = to16(dp[2 * s][r], elem());
            sf[4 + r] = to16(dp[2 * s + 1][r], elem());
        }
#pragma unroll
        for (int dt = 0; dt < 4; dt++) {
            dv[dt] = mfma16(get_cols<DT>(ldsDt, q0 + 32 * s, dt, g, c16), pf, dv[dt]);
            dk[dt] = mfma16(get_cols<DT>(ldsQt, q0 + 32 * s, dt, g, c16), sf, dk[dt]);
        }
    }
}

template <int DT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void attention_dkv_kernel(const BwdArgs a)
{
    typedef typename T16<DT>::elem elem;
    typedef typename T16<DT>::v8 v8;
    constexpr int THREADS = WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int S = a.S, W = a.W;
    const int SP = 32 * ((S + 31) / 32);
    const int CP = SP < CHUNK ? SP : CHUNK;
    unsigned char *ldsQ = smem, *ldsD = smem + CP * 128, *ldsQt = smem + 2 * CP * 128, *ldsDt = smem + 3 * CP * 128;
    float *lse_s = reinterpret_cast<float *>(smem + 4 * CP * 128), *delta_s = lse_s + CP;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c16 = lane & 15;
    const int head = blockIdx.x % a.heads, seq = blockIdx.x / a.heads;
    const long ld = 3L * W;
    const elem *base = (const elem *)a.qkv + (long)seq * S * ld + head * 64;
    const elem *dobase = (const elem *)a.dout + (long)seq * S * W + head * 64;
    const long stat = ((long)seq * a.heads + head) * S;

    const int n_kt = (S + 15) / 16;
    const int passes = (n_kt + WAVES - 1) / WAVES;
    for (int pass = 0; pass < passes; pass++) {
        const int kt = pass * WAVES + wave;
        const bool active = kt < n_kt;
        const int krow = kt * 16 + c16;
        const int ksrc = krow < S ? krow : S - 1;
        v8 kf[2], vf[2];
        if (active) {
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                kf[ks] = *reinterpret_cast<const v8 *>(base + (long)ksrc * ld + W + ks * 32 + g * 8);
                vf[ks] = *reinterpret_cast<const v8 *>(base + (long)ksrc * ld + 2 * W + ks * 32 + g * 8);
            }
        }
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int dt = 0; dt < 4; dt++) dk[dt] = dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int qbase = 0; qbase < SP; qbase += CP) {
            const int rows = SP - qbase < CP ? SP - qbase : CP;
            if (pass > 0 || qbase > 0) __syncthreads();
            if (pass == 0 || SP > CP) {
                const int r_in = threadIdx.x >> 3, ch = threadIdx.x & 7;
                for (int row = r_in; row < rows; row += THREADS / 8) {
                    const bool real = qbase + row < S;
                    const int srow = real ? qbase + row : S - 1;
                    const u32x4 qv = *reinterpret_cast<const u32x4 *>(base + (long)srow * ld + ch * 8);
                    u32x4 dv16 = *reinterpret_cast<const u32x4 *>(dobase + (long)srow * W + ch * 8);
                    if (!real) dv16 = u32x4{0u, 0u, 0u, 0u};
                    put_rows(ldsQ, row, ch, qv);
                    put_rows(ldsD, row, ch, dv16);
                    put_cols(ldsQt, row, ch, qv);
                    put_cols(ldsDt, row, ch, dv16);
                }
                for (int row = threadIdx.x; row < rows; row += THREADS) {
                    const bool real = qbase + row < S;
                    lse_s[row] = real ? a.lse[stat + qbase + row] : INFINITY;
                    delta_s[row] = real ? a.delta[stat + qbase + row] : 0.f;
                }
            }
            __syncthreads();
            if (active) {
                int q0 = 0;
                for (; q0 + 64 <= rows; q0 += 64)
                    dkv_block<DT, 2>(ldsQ, ldsD, ldsQt, ldsDt, lse_s, delta_s, q0, kf, vf, a.scale_log2e, dk, dv, g,
                                     c16);
                if (q0 < rows)
                    dkv_block<DT, 1>(ldsQ, ldsD, ldsQt, ldsDt, lse_s, delta_s, q0, kf, vf, a.scale_log2e, dk, dv, g,
                                     c16);
            }
        }
        if (active && krow < S) {
            elem kv[16], vv[16];
#pragma unroll
            for (int dt = 0; dt < 4; dt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    kv[4 * dt + r] = to16(dk[dt][r] * 0.125f, elem());
                    vv[4 * dt + r] = to16(dv[dt][r], elem());
                }
            elem *dst = (elem *)a.dqkv + ((long)seq * S + krow) * ld + head * 64 + g * 16;
            *reinterpret_cast<u32x4 *>(dst + W) = *reinterpret_cast<const u32x4 *>(&kv[0]);
            *reinterpret_cast<u32x4 *>(dst + W + 8) = *reinterpret_cast<const u32x4 *>(&kv[8]);
            *reinterpret_cast<u32x4 *>(dst + 2 * W) = *reinterpret_cast<const u32x4 *>(&vv[0]);
            *reinterpret_cast<u32x4 *>(dst + 2 * W + 8) = *reinterpret_cast<const u32x4 *>(&vv[8]);
        }
    }
}

template <int DT> int launch_bwd(const BwdArgs &a, int n_seq, hipStream_t s)
{
    const int SP = 32 * ((a.S + 31) / 32);
    const int CP = SP < CHUNK ? SP : CHUNK;
    const int lds_dq = 3 * CP * 128, lds_dkv = 4 * CP * 128 + 2 * CP * 4;
    // One workgroup per CU (the images fill the LDS), so the wave count decides how many passes the ceil(S / 16) row
    // tiles take: 17 tiles (S = 257) are three passes of 8 waves -- the last with one wave busy -- but two passes of 9.
    // Registers allow three waves per SIMD for the dQ body (133) and, held to 168, for the dK / dV body.
    const int n_t = (a.S + 15) / 16;
    const bool nine = lds_dq > 80 * 1024 && (n_t + 8) / 9 < (n_t + 7) / 8;
    void (*kq)(const BwdArgs) = nine ? attention_dq_kernel<DT, 9> : attention_dq_kernel<DT, 8>;
    void (*kkv)(const BwdArgs) = nine ? attention_dkv_kernel<DT, 9> : attention_dkv_kernel<DT, 8>;
    if (int rc = ec::ensure_dynamic_lds(reinterpret_cast<const void *>(kq), 160 * 1024)) return rc;
    if (int rc = ec::ensure_dynamic_lds(reinterpret_cast<const void *>(kkv), 160 * 1024)) return rc;
    const unsigned grid = (unsigned)a.heads * (unsigned)n_seq;
    // 3 + 4 score-sized products of 2 S^2 64 flops per head; reads q k v o dO, writes dq dk dv
    ec::ProfScope prof(ec::PROF_ATTENTION_BWD, s, 14.0 * a.S * a.S * 64.0 * a.heads * n_seq,
                       (double)n_seq * a.S * a.W * 2.0 * 8.0);
    hipLaunchKernelGGL(kq, dim3(grid), dim3(nine ? 576 : 512), lds_dq, s, a);
    hipLaunchKernelGGL(kkv, dim3(grid), dim3(nine ? 576 : 512), lds_dkv, s, a);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

}  // namespace

extern "C" EC_API int ec_attention_backward(const void *qkv, const void *out, const float *lse, const void *d_out,
                                            void *d_qkv, float *delta, int n_seq, int S, int width, int heads,
                                            int dtype, ec_stream_t stream)
{
    EC_REQUIRE(n_seq >= 0 && S > 0 && heads > 0, "ec_attention_backward: bad shape");
    EC_REQUIRE(width == heads * 64, "ec_attention_backward: head dim must be 64 (width %d, heads %d)", width,
               heads);
    if (n_seq == 0) return EC_OK;
    EC_REQUIRE(qkv && out && lse && d_out && d_qkv && delta, "ec_attention_backward: null buffer");
    BwdArgs a;
    a.qkv = qkv, a.out = out, a.dout = d_out, a.lse = lse, a.delta = delta, a.dqkv = d_qkv;
    a.S = S, a.W = width, a.heads = heads;
    a.scale_log2e = 0.125f * 1.4426950408889634f;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == EC_F16) return launch_bwd<EC_F16>(a, n_seq, s);
    if (dtype == EC_BF16) return launch_bwd<EC_BF16>(a, n_seq, s);
    return ec::fail(EC_ERR_INVALID, "ec_attention_backward: unknown dtype %d", dtype);
}
