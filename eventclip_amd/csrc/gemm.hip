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
};

__device__ __forceinline__ int swz_key(int row) { return (row & 7) ^ ((row >> 3) & 6); }

__device__ __forceinline__ float quick_gelu(float x)
{
    // QuickGELU of openai/CLIP: x * sigmoid(1.702 x)
    return x / (1.f + __expf(-1.702f * x));
}

// bijective XCD remap (blocks b and b+8 share an XCD): XCD x gets a contiguous id range
__device__ __forceinline__ int xcd_remap(int bid, int nblk)
{
    const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + (bid >> 3);
}

template <int DT, int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(WM *WN * 64) void gemm_kernel(const GemmArgs g)
{
    typedef typename T16<DT>::elem elem;
    typedef typename T16<DT>::v8 v8;
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM / 16;  // 16-row activation tiles per wave
    constexpr int TN = BN / WN / 16;  // 16-row weight tiles per wave
    static_assert(TN == 4, "epilogue assumes 64 output columns per wave");
    constexpr int PIECES = (BM + BN) / 8;  // 1-KiB DMA pieces (8 rows x 128 B) per stage
    constexpr int PPW = PIECES / NW;       // pieces per wave
    static_assert(PIECES % NW == 0, "stage must divide evenly over the waves");
    constexpr int STAGE_BYTES = (BM + BN) * 128;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int tile = xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
    const int m0 = (tile / g.tiles_n) * BM;
    const int n0 = (tile % g.tiles_n) * BN;

    // ---- per-lane DMA source pointers: piece p covers stage rows 8p .. 8p+7 ----
    const unsigned char *src[PPW];
#pragma unroll
    for (int i = 0; i < PPW; i++) {
        const int piece = i * NW + wave;
        const int row = piece * 8 + (lane >> 3);  // row within the stage (A rows, then W rows)
        const int chunk = (lane & 7) ^ swz_key(row);
        if (piece < BM / 8) {
            int m = m0 + row;
            m = m < g.M ? m : g.M - 1;
            src[i] = (const unsigned char *)g.A + ((long)m * g.lda + chunk * 8) * 2;
        } else {
            int n = n0 + (row - BM);
            n = n < g.N ? n : g.N - 1;
            src[i] = (const unsigned char *)g.W + ((long)n * g.K + chunk * 8) * 2;
        }
    }
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PPW; i++) {
            glds16(src[i], smem + buf * STAGE_BYTES + (i * NW + wave) * 1024);
            src[i] += BK * 2;
        }
    };

    // ---- per-lane fragment read offsets within a stage (k-substep 0) ----
    // activation rows (MFMA B operand): natural order
    int offB[TM];
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int row = wm * (BM / WM) + i * 16 + (lane & 15);
        offB[i] = row * 128 + (((lane >> 4) ^ swz_key(row)) << 4);
    }
    // weight rows (MFMA A operand): row i of tile j is output column
    // (i>>2)*16 + j*4 + (i&3) of this wave's 64, so that lane group g = lane>>4
    // owns columns 16g .. 16g+15 after the MFMAs.
    int offA[TN];
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int i = lane & 15;
        const int row = BM + wn * 64 + (i >> 2) * 16 + j * 4 + (i & 3);
        offA[j] = row * 128 + (((lane >> 4) ^ swz_key(row)) << 4);
    }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = g.K / BK;
    stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < nk; kt++) {
        if (kt + 1 < nk) stage(cur ^ 1);
        const unsigned char *sb = smem + cur * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            v8 fa[TN], fb[TM];
            // k-substep ks selects chunks 4ks..4ks+3: XOR bit 2 of the chunk = byte 64
#pragma unroll
            for (int j = 0; j < TN; j++)
                fa[j] = *reinterpret_cast<const v8 *>(sb + (offA[j] ^ (ks << 6)));
#pragma unroll
            for (int i = 0; i < TM; i++)
                fb[i] = *reinterpret_cast<const v8 *>(sb + (offB[i] ^ (ks << 6)));
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) acc[i][j] = mfma16(fa[j], fb[i], acc[i][j]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: lane owns row m = .. + (lane&15), columns nb .. nb+15 ----
    const int nb = n0 + wn * 64 + (lane >> 4) * 16;
    if (nb >= g.N) return;
    float bias[16];
#pragma unroll
    for (int c = 0; c < 16; c++) bias[c] = 0.f;
    if (g.bias) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const float4 b = *reinterpret_cast<const float4 *>(g.bias + nb + 4 * c);
            bias[4 * c] = b.x, bias[4 * c + 1] = b.y, bias[4 * c + 2] = b.z, bias[4 * c + 3] = b.w;
        }
    }
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int m = m0 + wm * (BM / WM) + i * 16 + (lane & 15);
        if (m >= g.M) continue;
        float v[16];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) v[4 * j + r] = acc[i][j][r] + bias[4 * j + r];
        if constexpr (EPI == EC_EPI_STORE16 || EPI == EC_EPI_GELU16) {
            elem o[16];
#pragma unroll
            for (int c = 0; c < 16; c++) {
                float x = v[c];
                if constexpr (EPI == EC_EPI_GELU16) x = quick_gelu(x);
                o[c] = to16(x, elem());
            }
            elem *dst = (elem *)g.C + (long)m * g.ldc + nb;
            *reinterpret_cast<u32x4 *>(dst) = *reinterpret_cast<const u32x4 *>(&o[0]);
            *reinterpret_cast<u32x4 *>(dst + 8) = *reinterpret_cast<const u32x4 *>(&o[8]);
        } else {
            float *dst = (float *)g.C + (long)m * g.ldc + nb;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float4 o = make_float4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
                if constexpr (EPI == EC_EPI_RESID32) {
                    const float4 x = *reinterpret_cast<const float4 *>(dst + 4 * c);
                    o.x += x.x, o.y += x.y, o.z += x.z, o.w += x.w;
                }
                *reinterpret_cast<float4 *>(dst + 4 * c) = o;
            }
        }
    }
}

template <int DT, int BM, int BN, int WM, int WN, int EPI>
int launch(const GemmArgs &g0, hipStream_t stream)
{
    GemmArgs g = g0;
    g.tiles_m = ec::ceil_div(g.M, BM);
    g.tiles_n = ec::ceil_div(g.N, BN);
    constexpr int lds = 2 * (BM + BN) * 128;
    auto kern = gemm_kernel<DT, BM, BN, WM, WN, EPI>;
    static bool attr_set = false;
    if (!attr_set) {
        EC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    constexpr int cls = EPI == EC_EPI_STORE16 ? ec::PROF_GEMM_STORE16
                        : EPI == EC_EPI_GELU16 ? ec::PROF_GEMM_GELU16
                        : EPI == EC_EPI_RESID32 ? ec::PROF_GEMM_RESID32 : ec::PROF_GEMM_STORE32;
    constexpr double out_b = (EPI == EC_EPI_STORE16 || EPI == EC_EPI_GELU16) ? 2.0
                             : (EPI == EC_EPI_RESID32 ? 8.0 : 4.0);
    ec::ProfScope prof(cls, stream, 2.0 * g.M * g.N * g.K,
                       2.0 * g.M * g.K + 2.0 * g.N * g.K + out_b * g.M * g.N);
    hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n), dim3(WM * WN * 64), lds, stream, g);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

template <int DT, int EPI> int dispatch_variant(const GemmArgs &g, int variant, hipStream_t s)
{
    switch (variant) {
    case 0:
    case 1: return launch<DT, 256, 256, 2, 4, EPI>(g, s);
    case 2: return launch<DT, 128, 128, 2, 2, EPI>(g, s);
    case 3: return launch<DT, 128, 256, 1, 4, EPI>(g, s);
    default: return ec::fail(EC_ERR_INVALID, "ec_gemm: unknown variant %d", variant);
    }
}

template <int DT> int dispatch_epi(const GemmArgs &g, int epi, int variant, hipStream_t s)
{
    switch (epi) {
    case EC_EPI_STORE16: return dispatch_variant<DT, EC_EPI_STORE16>(g, variant, s);
    case EC_EPI_GELU16: return dispatch_variant<DT, EC_EPI_GELU16>(g, variant, s);
    case EC_EPI_RESID32: return dispatch_variant<DT, EC_EPI_RESID32>(g, variant, s);
    case EC_EPI_STORE32: return dispatch_variant<DT, EC_EPI_STORE32>(g, variant, s);
    default: return ec::fail(EC_ERR_INVALID, "ec_gemm: unknown epilogue %d", epi);
    }
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
    EC_REQUIRE((((uintptr_t)a->A | (uintptr_t)a->W | (uintptr_t)a->C) & 15) == 0,
               "ec_gemm: buffers must be 16-byte aligned");
    GemmArgs g;
    g.M = a->M, g.N = a->N, g.K = a->K;
    g.A = a->A, g.lda = lda, g.W = a->W, g.bias = a->bias, g.C = a->C, g.ldc = ldc;
    g.tiles_m = g.tiles_n = 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (a->dtype == EC_F16) return dispatch_epi<EC_F16>(g, a->epilogue, a->variant, s);
    if (a->dtype == EC_BF16) return dispatch_epi<EC_BF16>(g, a->epilogue, a->variant, s);
    return ec::fail(EC_ERR_INVALID, "ec_gemm: unknown dtype %d", a->dtype);
}
