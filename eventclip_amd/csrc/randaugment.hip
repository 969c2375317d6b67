// RandAugment on uint8 event frames (gfx950): the 14 operators of the reference's
// datasets/augment.py (_apply_op :10-87, _augmentation_space :123-140), which the reference applies
// to PIL images through torchvision 0.13.1's functional_pil during few-shot / fine-tune training
// (datasets/event2img.py:36-42,120-121).  Every operator reproduces Pillow bit for bit:
//
//  * ShearX/Y, TranslateX/Y, Rotate -> Image.transform(AFFINE, BICUBIC, fillcolor): libImaging
//    Geometry.c affine_transform + bicubic_filter32RGB, float64, a = -1 cubic, clamped neighbours,
//    (UINT8) truncation, untouched (= fill colour) outside the source.  The 6 coefficients are
//    computed on the host exactly as torchvision / PIL compute them.
//  * Brightness / Color / Contrast / Sharpness -> ImageEnhance = Image.blend(degenerate, image, f):
//    Blend.c in C float; degenerate = black / L(pixel) / the frame's mean L / the 3x3 SMOOTH filter
//    (Filter.c, float, borders copied).
//  * Posterize / Solarize / AutoContrast / Equalize -> 256-entry per-band look-up tables (ImageOps),
//    the last two from the frame's own per-band histograms.
//
// HBM-bound byte work: one pass over the frame per operator (plus a histogram pass for the three
// operators that need frame statistics), no MFMA.  FP contraction is off: Pillow's arithmetic has
// one rounding per operation.
#include "common.h"

#pragma clang fp contract(off)

namespace {

constexpr int AUG_THREADS = 256;
constexpr int STATS_WORDS = 3 * 256 + 2;   // per frame: band histograms, then the 64-bit sum of L

__device__ __forceinline__ int to_L(int r, int g, int b)
{
    return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16;   // Pillow's RGB -> L
}

__device__ __forceinline__ bool needs_stats(int kind)
{
    return kind == EC_AUG_CONTRAST || kind == EC_AUG_AUTOCONTRAST || kind == EC_AUG_EQUALIZE;
}

// per-frame statistics of the step's input: histogram of every band and the sum of L
__global__ __launch_bounds__(AUG_THREADS) void aug_stats_kernel(const uint8_t *in, const ec_aug_op *ops,
                                                                int step, int num_ops, int H, int W,
                                                                int bands, unsigned *stats)
{
    const int f = blockIdx.x / bands, band = blockIdx.x % bands;
    const int kind = ops[(long)f * num_ops + step].kind;
    if (!needs_stats(kind)) return;
    __shared__ unsigned hist[3 * 256];
    __shared__ unsigned long long lsum;
    for (int i = threadIdx.x; i < 3 * 256; i += AUG_THREADS) hist[i] = 0;
    if (threadIdx.x == 0) lsum = 0;
    __syncthreads();
    const long npix = (long)H * W;
    const long per = (npix + bands - 1) / bands;
    const long p0 = band * per, p1 = p0 + per < npix ? p0 + per : npix;
    const uint8_t *src = in + (long)f * npix * 3;
    unsigned long long mine = 0;
    for (long p = p0 + threadIdx.x; p < p1; p += AUG_THREADS) {
        const int r = src[3 * p], g = src[3 * p + 1], b = src[3 * p + 2];
        if (kind == EC_AUG_CONTRAST) {
            mine += (unsigned)to_L(r, g, b);
        } else {
            atomicAdd(&hist[r], 1u);
            atomicAdd(&hist[256 + g], 1u);
            atomicAdd(&hist[512 + b], 1u);
        }
    }
    unsigned *st = stats + (long)f * STATS_WORDS;
    if (kind == EC_AUG_CONTRAST) {
        atomicAdd(&lsum, mine);
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long *>(st + 3 * 256), lsum);
    } else {
        __syncthreads();
        for (int i = threadIdx.x; i < 3 * 256; i += AUG_THREADS)
            if (hist[i]) atomicAdd(&st[i], hist[i]);
    }
}

// Geometry.c: BICUBIC(v, v1, v2, v3, v4, d)
__device__ __forceinline__ double cubic(double v1, double v2, double v3, double v4, double d)
{
    const double p1 = v2;
    const double p2 = -v1 + v3;
    const double p3 = 2 * (v1 - v2) + v3 - v4;
    const double p4 = -v1 + v2 - v3 + v4;
    return p1 + d * (p2 + d * (p3 + d * p4));
}

__device__ __forceinline__ uint8_t clip8_trunc(double v)
{
    return v <= 0.0 ? (uint8_t)0 : (v >= 255.0 ? (uint8_t)255 : (uint8_t)(int)v);
}

// Blend.c: in1 + alpha * (in2 - in1) in C float, truncated; clipped when alpha extrapolates
__device__ __forceinline__ uint8_t blend8(int deg, int img, float alpha, bool inside01)
{
    const float t = (float)deg + alpha * (float)(img - deg);
    if (inside01) return (uint8_t)(int)t;
    return t <= 0.f ? (uint8_t)0 : (t >= 255.f ? (uint8_t)255 : (uint8_t)(int)t);
}

__global__ __launch_bounds__(AUG_THREADS) void aug_apply_kernel(const uint8_t *in, uint8_t *out,
                                                                const ec_aug_op *ops, int step,
                                                                int num_ops, int H, int W, int bands,
                                                                const unsigned *stats, uint8_t fr,
                                                                uint8_t fg, uint8_t fb)
{
    const int f = blockIdx.x / bands, band = blockIdx.x % bands;
    const ec_aug_op op = ops[(long)f * num_ops + step];
    const long npix = (long)H * W;
    const long per = ((npix + bands - 1) / bands + 3) / 4 * 4;        // bands start on multiples of four pixels
    const long p0 = band * per < npix ? band * per : npix, p1 = p0 + per < npix ? p0 + per : npix;
    const uint8_t *src = in + (long)f * npix * 3;
    uint8_t *dst = out + (long)f * npix * 3;
    const int kind = op.kind;
    const bool quads = ((npix * 3) & 3) == 0 && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 3) == 0;

    __shared__ uint8_t lut[3 * 256];
    __shared__ int mean_l;
    if (kind == EC_AUG_POSTERIZE || kind == EC_AUG_SOLARIZE) {
        for (int i = threadIdx.x; i < 256; i += AUG_THREADS) {
            uint8_t v;
            if (kind == EC_AUG_POSTERIZE) {
                const int mask = ~((1 << (8 - (int)op.param)) - 1);       // ImageOps.posterize
                v = (uint8_t)(i & mask);
            } else {
                v = (double)i < op.param ? (uint8_t)i : (uint8_t)(255 - i);   // ImageOps.solarize
            }
            lut[i] = lut[256 + i] = lut[512 + i] = v;
        }
    } else if (kind == EC_AUG_AUTOCONTRAST) {
        const unsigned *st = stats + (long)f * STATS_WORDS;
        if (threadIdx.x < 3) {
            const unsigned *h = st + threadIdx.x * 256;
            int lo = 0, hi = 255;
            while (lo < 256 && !h[lo]) lo++;
            while (hi >= 0 && !h[hi]) hi--;
            uint8_t *l = lut + threadIdx.x * 256;
            if (hi <= lo) {
                for (int i = 0; i < 256; i++) l[i] = (uint8_t)i;
            } else {
                const double scale = 255.0 / (double)(hi - lo);
                const double offset = -(double)lo * scale;
                for (int i = 0; i < 256; i++) {
                    const double t = (double)i * scale + offset;
                    const int v = (int)t;                                  // python int(): towards zero
                    l[i] = v < 0 ? (uint8_t)0 : (v > 255 ? (uint8_t)255 : (uint8_t)v);
                }
            }
        }
    } else if (kind == EC_AUG_EQUALIZE) {
        const unsigned *st = stats + (long)f * STATS_WORDS;
        if (threadIdx.x < 3) {
            const unsigned *h = st + threadIdx.x * 256;
            uint8_t *l = lut + threadIdx.x * 256;
            long total = 0, last = 0;
            int nonzero = 0;
            for (int i = 0; i < 256; i++)
                if (h[i]) total += h[i], last = h[i], nonzero++;
            const long stepv = nonzero <= 1 ? 0 : (total - last) / 255;
            if (!stepv) {
                for (int i = 0; i < 256; i++) l[i] = (uint8_t)i;
            } else {
                long n = stepv / 2;
                for (int i = 0; i < 256; i++) {
                    const long v = n / stepv;
                    l[i] = v > 255 ? (uint8_t)255 : (uint8_t)v;           // Image.point clips the table
                    n += h[i];
                }
            }
        }
    } else if (kind == EC_AUG_CONTRAST) {
        if (threadIdx.x == 0) {
            const unsigned long long s =
                *reinterpret_cast<const unsigned long long *>(stats + (long)f * STATS_WORDS + 3 * 256);
            mean_l = (int)((double)s / (double)npix + 0.5);                // ImageStat mean + 0.5, int()
        }
    }
    __syncthreads();

    const float alpha = op.alpha;
    const bool inside01 = alpha >= 0.f && alpha <= 1.f;
    const uint8_t fillc[3] = {fr, fg, fb};
    // One output pixel.  v: the pixel's own source bytes (the operators that are functions of them get them from
    // the caller, who reads four pixels as three dwords).
    auto pixel = [&](long p, const int (&v)[3], uint8_t (&o)[3]) {
        const int y = (int)(p / W), x = (int)(p - (long)y * W);
        switch (kind) {
        case EC_AUG_AFFINE: {
            const double xs = (double)x + 0.5, ys = (double)y + 0.5;
            double xin = op.m[0] * xs + op.m[1] * ys + op.m[2];
            double yin = op.m[3] * xs + op.m[4] * ys + op.m[5];
            if (xin < 0.0 || xin >= (double)W || yin < 0.0 || yin >= (double)H) {
                o[0] = fillc[0], o[1] = fillc[1], o[2] = fillc[2];
                break;
            }
            xin -= 0.5;
            yin -= 0.5;
            const double fx = floor(xin), fy = floor(yin);
            const double dx = xin - fx, dy = yin - fy;
            const int x0 = (int)fx - 1, y0 = (int)fy - 1;
            int xc[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int xx = x0 + k;
                xc[k] = xx < 0 ? 0 : (xx < W ? xx : W - 1);
            }
            // the four taps of a row are 12 contiguous bytes away from the left / right borders: three unaligned
            // dwords instead of twelve byte loads (the kernel ran at the texture unit's instruction rate)
            const bool contiguous = x0 >= 0 && x0 + 3 < W;
            double rows[3][4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int yy = y0 + k;
                if (k == 0 || (yy >= 0 && yy < H)) {
                    const int yr = yy < 0 ? 0 : (yy < H ? yy : H - 1);   // YCLIP (first row only)
                    const uint8_t *r = src + ((long)yr * W) * 3;
                    int t[4][3];
                    if (contiguous) {
                        unsigned w[3];
                        __builtin_memcpy(w, r + (long)x0 * 3, 12);
#pragma unroll
                        for (int i = 0; i < 12; i++) t[i / 3][i % 3] = (int)((w[i >> 2] >> (8 * (i & 3))) & 255u);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; i++)
#pragma unroll
                            for (int c = 0; c < 3; c++) t[i][c] = r[xc[i] * 3 + c];
                    }
#pragma unroll
                    for (int c = 0; c < 3; c++)
                        rows[c][k] = cubic((double)t[0][c], (double)t[1][c], (double)t[2][c], (double)t[3][c], dx);
                } else {
#pragma unroll
                    for (int c = 0; c < 3; c++) rows[c][k] = rows[c][k - 1];
                }
            }
#pragma unroll
            for (int c = 0; c < 3; c++) o[c] = clip8_trunc(cubic(rows[c][0], rows[c][1], rows[c][2], rows[c][3], dy));
            break;
        }
        case EC_AUG_ROT180: {
            const uint8_t *s = src + ((long)(H - 1 - y) * W + (W - 1 - x)) * 3;
            o[0] = s[0], o[1] = s[1], o[2] = s[2];
            break;
        }
        case EC_AUG_ROT90: {    // Image.ROTATE_90 on a square frame (counter-clockwise)
            const uint8_t *s = src + ((long)x * W + (W - 1 - y)) * 3;
            o[0] = s[0], o[1] = s[1], o[2] = s[2];
            break;
        }
        case EC_AUG_ROT270: {
            const uint8_t *s = src + ((long)(H - 1 - x) * W + y) * 3;
            o[0] = s[0], o[1] = s[1], o[2] = s[2];
            break;
        }
        case EC_AUG_BRIGHTNESS:
        case EC_AUG_COLOR:
        case EC_AUG_CONTRAST:
        case EC_AUG_SHARPNESS: {
            if (alpha == 1.f) {                      // Image.blend returns a copy of the image
                o[0] = (uint8_t)v[0], o[1] = (uint8_t)v[1], o[2] = (uint8_t)v[2];
                break;
            }
            int deg[3];
            if (kind == EC_AUG_BRIGHTNESS) {
                deg[0] = deg[1] = deg[2] = 0;
            } else if (kind == EC_AUG_COLOR) {
                deg[0] = deg[1] = deg[2] = to_L(v[0], v[1], v[2]);
            } else if (kind == EC_AUG_CONTRAST) {
                deg[0] = deg[1] = deg[2] = mean_l;
            } else if (x == 0 || y == 0 || x == W - 1 || y == H - 1) {
                deg[0] = v[0], deg[1] = v[1], deg[2] = v[2];      // Filter.c copies the border
            } else {
                // SMOOTH: (1 1 1 / 1 5 1 / 1 1 1) / 13 in float, rows y+1, y, y-1, then + 0.5, truncate
                const float k1 = 1.f / 13.f, k5 = 5.f / 13.f;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const uint8_t *a = src + ((long)(y + 1) * W + x) * 3 + c;
                    const uint8_t *b = src + ((long)y * W + x) * 3 + c;
                    const uint8_t *d = src + ((long)(y - 1) * W + x) * 3 + c;
                    float ss = 0.5f;
                    ss += (float)a[-3] * k1 + (float)a[0] * k1 + (float)a[3] * k1;
                    ss += (float)b[-3] * k1 + (float)b[0] * k5 + (float)b[3] * k1;
                    ss += (float)d[-3] * k1 + (float)d[0] * k1 + (float)d[3] * k1;
                    deg[c] = ss <= 0.f ? 0 : (ss >= 255.f ? 255 : (int)ss);
                }
            }
            if (alpha == 0.f) {
                o[0] = (uint8_t)deg[0], o[1] = (uint8_t)deg[1], o[2] = (uint8_t)deg[2];
            } else {
#pragma unroll
                for (int c = 0; c < 3; c++) o[c] = blend8(deg[c], v[c], alpha, inside01);
            }
            break;
        }
        case EC_AUG_POSTERIZE:
        case EC_AUG_SOLARIZE:
        case EC_AUG_AUTOCONTRAST:
        case EC_AUG_EQUALIZE: {
            o[0] = lut[v[0]], o[1] = lut[256 + v[1]], o[2] = lut[512 + v[2]];
            break;
        }
        default: {   // EC_AUG_IDENTITY
            o[0] = (uint8_t)v[0], o[1] = (uint8_t)v[1], o[2] = (uint8_t)v[2];
        }
        }
    };
    const bool own_bytes = !(kind == EC_AUG_AFFINE || kind == EC_AUG_ROT180 || kind == EC_AUG_ROT90 || kind == EC_AUG_ROT270);
    // four pixels = twelve bytes = three dwords per thread, in (the operators above that want them) and out; a
    // band starts on a multiple of four pixels and a frame's byte size is a multiple of four when `quads`
    long p_tail = p0;
    if (quads) {
        const long g1 = p1 / 4;
        for (long g = p0 / 4 + threadIdx.x; g < g1; g += AUG_THREADS) {
            unsigned in[3] = {0, 0, 0}, out3[3] = {0, 0, 0};
            if (own_bytes) {
                const unsigned *s4 = reinterpret_cast<const unsigned *>(src + g * 12);
                in[0] = s4[0], in[1] = s4[1], in[2] = s4[2];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                int v[3];
                uint8_t o[3];
#pragma unroll
                for (int c = 0; c < 3; c++) v[c] = (int)((in[(3 * j + c) >> 2] >> (8 * ((3 * j + c) & 3))) & 255u);
                pixel(g * 4 + j, v, o);
#pragma unroll
                for (int c = 0; c < 3; c++) out3[(3 * j + c) >> 2] |= (unsigned)o[c] << (8 * ((3 * j + c) & 3));
            }
            unsigned *d4 = reinterpret_cast<unsigned *>(dst + g * 12);
            d4[0] = out3[0], d4[1] = out3[1], d4[2] = out3[2];
        }
        p_tail = g1 * 4 > p0 ? g1 * 4 : p0;
    }
    for (long p = p_tail + threadIdx.x; p < p1; p += AUG_THREADS) {
        const uint8_t *s = src + p * 3;
        int v[3] = {0, 0, 0};
        if (own_bytes) v[0] = s[0], v[1] = s[1], v[2] = s[2];
        uint8_t o[3];
        pixel(p, v, o);
        dst[p * 3] = o[0], dst[p * 3 + 1] = o[1], dst[p * 3 + 2] = o[2];
    }
}

}  // namespace

extern "C" {

EC_API size_t ec_randaugment_workspace_bytes(int F, int H, int W, int num_ops)
{
    if (F <= 0 || H <= 0 || W <= 0 || num_ops <= 0) return 0;
    // one intermediate frame set (ping-pong for num_ops > 1) + per-frame statistics
    const size_t frames = num_ops > 1 ? (size_t)F * H * W * 3 : 0;
    return ((frames + 255) & ~(size_t)255) + (size_t)F * STATS_WORDS * 4;
}

EC_API int ec_randaugment(const uint8_t *frames_in, uint8_t *frames_out, int F, int H, int W,
                          const ec_aug_op *ops, int num_ops, const uint8_t fill[3], void *workspace,
                          size_t workspace_bytes, ec_stream_t stream)
{
    EC_REQUIRE(F >= 0 && H > 0 && W > 0 && num_ops >= 1, "ec_randaugment: bad shape");
    if (F == 0) return EC_OK;
    EC_REQUIRE(frames_in && frames_out && ops && fill && workspace, "ec_randaugment: null buffer");
    EC_REQUIRE(frames_in != frames_out, "ec_randaugment: in place is not supported");
    const size_t need = ec_randaugment_workspace_bytes(F, H, W, num_ops);
    if (workspace_bytes < need)
        return ec::fail(EC_ERR_WORKSPACE, "ec_randaugment: workspace %zu < %zu bytes", workspace_bytes, need);
    const size_t frame_bytes = (size_t)F * H * W * 3;
    uint8_t *tmp = static_cast<uint8_t *>(workspace);
    unsigned *stats = reinterpret_cast<unsigned *>(tmp + (num_ops > 1 ? ((frame_bytes + 255) & ~(size_t)255) : 0));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int bands = ec::ceil_div(H * W, 16384);       // ~16 k pixels per workgroup
    const dim3 grid((unsigned)F * bands), block(AUG_THREADS);
    // step k reads what step k-1 wrote; the last step writes frames_out
    const uint8_t *src = frames_in;
    for (int k = 0; k < num_ops; k++) {
        uint8_t *dst = (k == num_ops - 1) ? frames_out : ((num_ops - 1 - k) % 2 ? tmp : frames_out);
        EC_CHECK_HIP(hipMemsetAsync(stats, 0, (size_t)F * STATS_WORDS * 4, s));
        hipLaunchKernelGGL(aug_stats_kernel, grid, block, 0, s, src, ops, k, num_ops, H, W, bands, stats);
        hipLaunchKernelGGL(aug_apply_kernel, grid, block, 0, s, src, dst, ops, k, num_ops, H, W, bands,
                           stats, fill[0], fill[1], fill[2]);
        EC_CHECK_HIP(hipGetLastError());
        src = dst;
    }
    return EC_OK;
}

}  // extern "C"
