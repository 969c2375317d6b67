// Fused multi-head self-attention for the CLIP towers (gfx950), head dim 64.
//
// Replaces nn.MultiheadAttention inside the ResidualAttentionBlocks of
// un-vendored openai/CLIP clip/model.py (reached through
// models/clip_cls.py:84 encode_text / :101 encode_image).  Sequences are short
// and fixed (50..577 vision tokens, 77 text tokens), so one workgroup owns one
// (sequence, head): the head's whole K and V (<= 608 x 64 x 16 bit each) sit in
// LDS, every wave takes 16-query tiles, and the full score row lives in
// registers -- no online-softmax rescaling is needed.
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
#include "common.h"
#include "mfma.h"

namespace {

using namespace ec;

struct AttnArgs {
    const void *qkv;  // [n_seq * S, 3W] 16-bit: q | k | v, heads are 64-wide column blocks
    void *out;        // [n_seq * S, W] 16-bit
    int S, W, heads, causal;
    float scale_log2e;
};

__device__ __forceinline__ float xor_max(float v)
{
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float xor_sum(float v)
{
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

template <int DT, int NT2>  // NT2 = number of 32-key steps; keys padded to 32 * NT2
__global__ __launch_bounds__(256) void attention_kernel(const AttnArgs a)
{
    typedef typename T16<DT>::elem elem;
    typedef typename T16<DT>::v8 v8;
    typedef typename T16<DT>::v4 v4;
    constexpr int NT = 2 * NT2;    // 16-key score tiles
    constexpr int SP = 32 * NT2;   // padded key count

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *ldsK = smem;
    unsigned char *ldsV = smem + SP * 128;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c16 = lane & 15;
    const int head = blockIdx.x % a.heads, seq = blockIdx.x / a.heads;
    const int S = a.S, W = a.W;
    const long ld = 3L * W;
    const elem *base = (const elem *)a.qkv + (long)seq * S * ld + head * 64;

    // ---- stage K and V of this head: 32 rows x 8 chunks of 16 B per pass ----
    {
        const int r_in = threadIdx.x >> 3, ch = threadIdx.x & 7;
#pragma unroll 2
        for (int it = 0; it < NT2; it++) {
            const int row = it * 32 + r_in;
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
    __syncthreads();

    const int n_qt = (S + 15) / 16;
    for (int qt = wave; qt < n_qt; qt += 4) {
        // ---- Q tile as the MFMA B operand: lane -> query c16, d = 32 ks + 8 g + j ----
        const int qrow = qt * 16 + c16;
        const int qsrc = qrow < S ? qrow : S - 1;
        v8 qf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
            qf[ks] = *reinterpret_cast<const v8 *>(base + (long)qsrc * ld + ks * 32 + g * 8);

        // ---- scores: acc[kt][r] = <k[16 kt + 4 g + r], q[c16]> ----
        f32x4 acc[NT];
#pragma unroll
        for (int kt = 0; kt < NT; kt++) {
            acc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int row = kt * 16 + c16;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const v8 kf = *reinterpret_cast<const v8 *>(
                    ldsK + row * 128 + (((ks * 4 + g) ^ (row & 7)) << 4));
                acc[kt] = mfma16(kf, qf[ks], acc[kt]);
            }
        }

        // ---- mask + softmax over keys (fp32) ----
        const int klimit = a.causal ? (qrow < S ? qrow + 1 : S) : S;  // keys < klimit are visible
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; kt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int key = kt * 16 + 4 * g + r;
                const float s = key < klimit ? acc[kt][r] : -INFINITY;
                acc[kt][r] = s;
                mx = fmaxf(mx, s);
            }
        mx = xor_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; kt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float p = __builtin_amdgcn_exp2f((acc[kt][r] - mx) * a.scale_log2e);
                acc[kt][r] = p;
                sum += p;
            }
        sum = xor_sum(sum);

        // ---- O^T = V^T . P^T over 32-key steps ----
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; dt++) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NT2; s++) {
            // B operand element j <-> key 32 s + 16 (j >> 2) + 4 g + (j & 3)
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
                    const int row = 32 * s + 16 * hh + 4 * g + (c16 >> 2);
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

        // ---- normalise and store: lane owns query c16, head dims 16 g .. 16 g + 15 ----
        if (qrow < S) {
            const float inv = 1.f / sum;
            elem ov[16];
#pragma unroll
            for (int dt = 0; dt < 4; dt++)
#pragma unroll
                for (int r = 0; r < 4; r++) ov[4 * dt + r] = to16(o[dt][r] * inv, elem());
            elem *dst = (elem *)a.out + ((long)seq * S + qrow) * W + head * 64 + g * 16;
            *reinterpret_cast<u32x4 *>(dst) = *reinterpret_cast<const u32x4 *>(&ov[0]);
            *reinterpret_cast<u32x4 *>(dst + 8) = *reinterpret_cast<const u32x4 *>(&ov[8]);
        }
    }
}

template <int DT, int NT2> int launch(const AttnArgs &a, int n_seq, int heads, hipStream_t s)
{
    constexpr int lds = 32 * NT2 * 128 * 2;
    auto kern = attention_kernel<DT, NT2>;
    static bool attr_set = false;
    if (!attr_set) {
        EC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    // algorithmic work: QK^T and PV, 2 * 2 * S^2 * 64 flops per head; bytes: read qkv, write out
    ec::ProfScope prof(ec::PROF_ATTENTION, s, 4.0 * a.S * a.S * 64.0 * heads * n_seq,
                       (double)n_seq * a.S * a.W * 2.0 * 4.0);
    hipLaunchKernelGGL(kern, dim3((unsigned)heads * (unsigned)n_seq), dim3(256), lds, s, a);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

template <int DT> int dispatch(const AttnArgs &a, int n_seq, int heads, hipStream_t s)
{
    const int nt2 = (a.S + 31) / 32;
    if (nt2 <= 2) return launch<DT, 2>(a, n_seq, heads, s);    // S <= 64  (ViT-B/32: 50)
    if (nt2 <= 3) return launch<DT, 3>(a, n_seq, heads, s);    // S <= 96  (text: 77)
    if (nt2 <= 7) return launch<DT, 7>(a, n_seq, heads, s);    // S <= 224 (ViT-B/16: 197)
    if (nt2 <= 9) return launch<DT, 9>(a, n_seq, heads, s);    // S <= 288 (ViT-L/14: 257)
    if (nt2 <= 19) return launch<DT, 19>(a, n_seq, heads, s);  // S <= 608 (ViT-L/14@336: 577)
    return ec::fail(EC_ERR_UNSUPPORTED, "ec_attention: sequence length %d > 608", a.S);
}

}  // namespace

extern "C" EC_API int ec_attention(const void *qkv, void *out, int n_seq, int S, int width,
                                   int heads, int causal, int dtype, ec_stream_t stream)
{
    EC_REQUIRE(n_seq >= 0 && S > 0 && heads > 0, "ec_attention: bad shape");
    EC_REQUIRE(width == heads * 64, "ec_attention: head dim must be 64 (width %d, heads %d)", width,
               heads);
    if (n_seq == 0) return EC_OK;
    EC_REQUIRE(qkv && out, "ec_attention: null buffer");
    AttnArgs a;
    a.qkv = qkv, a.out = out, a.S = S, a.W = width, a.heads = heads, a.causal = causal;
    a.scale_log2e = 0.125f * 1.4426950408889634f;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == EC_F16) return dispatch<EC_F16>(a, n_seq, heads, s);
    if (dtype == EC_BF16) return dispatch<EC_BF16>(a, n_seq, heads, s);
    return ec::fail(EC_ERR_INVALID, "ec_attention: unknown dtype %d", dtype);
}
