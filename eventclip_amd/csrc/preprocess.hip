// CLIP image preprocess on gfx950: uint8 frames -> normalised model input.
//
// Replaces the per-frame `self.transforms(img)` of the reference
// (datasets/event2img.py:119-122 with params.data_transforms = clip.load's
// preprocess, test.py:26-29): torchvision Resize(n_px, BICUBIC) ->
// CenterCrop(n_px) -> ToTensor -> Normalize, i.e. Pillow's 8-bit two-pass
// bicubic (Resample.c: double coefficients -> 22-bit fixed point, uint8
// rounding between the horizontal and the vertical pass), integer-exact here.
//
// One workgroup produces a band of whole patch rows of one frame: the
// horizontally resampled input rows the band needs are built in LDS (uint8),
// the vertical pass reads them back, the (x/255 - mean)/std step is a 3x256
// fp32 table computed on the host in torch's operation order, and the result
// is written either as fp32 CHW (the reference's tensor) or straight into the
// 16-bit im2col rows [F, G, kpad] the patch-embedding GEMM consumes (padding
// columns zeroed here), so the frame never exists in fp32 in HBM.
#include "common.h"
#include "mfma.h"

#include <cmath>
#include <type_traits>
#include <cstring>
#include <vector>

namespace {

using namespace ec;

constexpr int PRECISION_BITS = 32 - 8 - 2;
constexpr int PLAN_MAGIC = 0x45435031;  // "ECP1"
enum { P_MAGIC, P_IN_H, P_IN_W, P_NPX, P_NEW_H, P_NEW_W, P_TOP, P_LEFT, P_KS_H, P_KS_V, P_OFF_BH,
       P_OFF_KH, P_OFF_BV, P_OFF_KV, P_OFF_LUT, P_WORDS, P_HEADER = 16 };

double bicubic(double x)
{
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

int ksize_for(int in_size, int out_size)
{
    double fs = (double)in_size / out_size;
    if (fs < 1.0) fs = 1.0;
    return (int)std::ceil(2.0 * fs) * 2 + 1;
}

// Pillow precompute_coeffs + normalize_coeffs_8bpc for output positions
// first .. first + count - 1 of a full-box resample in_size -> out_size.
void coeffs(int in_size, int out_size, int first, int count, int ksize, int32_t *bounds, int32_t *kk)
{
    double scale = (double)in_size / out_size, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale, ss = 1.0 / filterscale;
    std::vector<double> w((size_t)ksize);
    for (int o = 0; o < count; o++) {
        const int xx = first + o;
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; x++) {
            w[x] = bicubic((x + xmin - center + 0.5) * ss);
            ww += w[x];
        }
        for (int x = 0; x < ksize; x++) {
            double v = 0.0;
            if (x < xmax) v = (ww != 0.0) ? w[x] / ww : w[x];
            kk[o * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS))
                                      : (int)(0.5 + v * (1 << PRECISION_BITS));
        }
        bounds[2 * o] = xmin;
        bounds[2 * o + 1] = xmax;
    }
}

void resized_size(int h, int w, int n_px, int *nh, int *nw)
{
    // torchvision Resize(int): short side -> n_px, long side = int(n_px * long / short)
    if (w <= h) {
        *nw = n_px;
        *nh = (int)((double)n_px * h / w);
    } else {
        *nh = n_px;
        *nw = (int)((double)n_px * w / h);
    }
}

int round_half_even_div2(int d)  // int(round(d / 2.0)) with Python's round
{
    if (d % 2 == 0) return d / 2;
    const int lo = (d - 1) / 2;  // d/2 = lo + 0.5 -> nearest even
    return (lo % 2 == 0) ? lo : lo + 1;
}

struct PreArgs {
    const uint8_t *frames;
    const int32_t *plan;
    void *out;
    int in_h, in_w, n_px, band_rows, bands, ks_h, ks_v, off_bh, off_kh, off_bv, off_kv, off_lut;
    int tmp_bytes;      // LDS bytes of the horizontally resampled rows; the table and the band's coefficients follow
    int mode, patch, kpad, grid_w;
};

template <int DT>
__global__ __launch_bounds__(256) void preprocess_kernel(const PreArgs a)
{
    typedef typename T16<DT>::elem elem;
    extern __shared__ __attribute__((aligned(16))) unsigned char tmp[];  // [ny][n_px][3]
    const int f = blockIdx.x / a.bands, band = blockIdx.x % a.bands;
    const int R = a.n_px;
    const int oy0 = band * a.band_rows;
    const int oy1 = min(R, oy0 + a.band_rows);
    const int32_t *bh = a.plan + a.off_bh, *kh = a.plan + a.off_kh;
    const int32_t *bv = a.plan + a.off_bv, *kv = a.plan + a.off_kv;
    // the normalisation table and this band's vertical coefficients go to LDS once (three table look-ups and
    // up to 11 coefficient loads per output pixel were texture-unit instructions)
    float *lut = reinterpret_cast<float *>(tmp + a.tmp_bytes);
    int32_t *kvs = reinterpret_cast<int32_t *>(tmp + a.tmp_bytes + 3 * 256 * 4);
    {
        const float *lut_g = reinterpret_cast<const float *>(a.plan + a.off_lut);
        for (int i = threadIdx.x; i < 3 * 256; i += 256) lut[i] = lut_g[i];
        const int nk = (oy1 - oy0) * a.ks_v;
        for (int i = threadIdx.x; i < nk; i += 256) kvs[i] = kv[oy0 * a.ks_v + i];
    }
    const int ymin = bv[2 * oy0];
    const int ymax = bv[2 * (oy1 - 1)] + bv[2 * (oy1 - 1) + 1];
    const int ny = ymax - ymin;
    const uint8_t *src = a.frames + (long)f * a.in_h * a.in_w * 3;

    // ---- horizontal pass into LDS ----
    // An item is an output column and every HG-th input row: its (at most 12) coefficients stay in registers, and
    // a row's tap window -- 3 cnt contiguous bytes -- is read as unaligned dwords, not byte by byte.  (One output
    // per item with byte loads issued 40 loads per output on N-ImageNet frames; the pass ran at the texture
    // unit's instruction rate: 6.3 ms per 2560 frames.)
    // (The tap count is a compile-time bound of the item's loop: 6 covers every upscale -- a bicubic window is 4 or 5
    // input pixels wide there, N-Caltech's 180 x 240 and N-Cars' frames -- and 12 the downscales to 2.75 x, N-ImageNet;
    // with one bound of 12 an upscale multiplied seven zero coefficients per output.)
    constexpr int HG = 8;
    const bool fast_h = a.ks_h <= 12;
    auto horizontal = [&](auto taps) {
        constexpr int HT = decltype(taps)::value, HW = (3 * HT + 3) / 4;
        for (int it = threadIdx.x; it < R * HG; it += 256) {
            const int ox = it % R, rg = it / R;
            const int xmin = bh[2 * ox], cnt = bh[2 * ox + 1];
            int coef[HT];
#pragma unroll
            for (int t = 0; t < HT; t++) coef[t] = t < cnt ? kh[ox * a.ks_h + t] : 0;
            const int nbytes = 3 * cnt, ndw = nbytes >> 2;
            for (int yy = rg; yy < ny; yy += HG) {
                const uint8_t *row = src + ((long)(ymin + yy) * a.in_w + xmin) * 3;
                unsigned w[HW];
#pragma unroll
                for (int j = 0; j < HW; j++) {
                    w[j] = 0;
                    if (j < ndw) {
                        unsigned v;
                        __builtin_memcpy(&v, row + 4 * j, 4);          // (unaligned global load)
                        w[j] = v;
                    } else if (j == ndw) {                             // the window's last 1-3 bytes, one at a time
                        for (int b = 0; b < (nbytes & 3); b++) w[j] |= (unsigned)row[4 * j + b] << (8 * b);
                    }
                }
                int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
#pragma unroll
                for (int t = 0; t < HT; t++) {
                    const int c = coef[t];
                    s0 += (int)((w[(3 * t) >> 2] >> (8 * ((3 * t) & 3))) & 255u) * c;
                    s1 += (int)((w[(3 * t + 1) >> 2] >> (8 * ((3 * t + 1) & 3))) & 255u) * c;
                    s2 += (int)((w[(3 * t + 2) >> 2] >> (8 * ((3 * t + 2) & 3))) & 255u) * c;
                }
                unsigned char *d = tmp + (yy * R + ox) * 3;
                d[0] = (unsigned char)min(255, max(0, s0 >> PRECISION_BITS));
                d[1] = (unsigned char)min(255, max(0, s1 >> PRECISION_BITS));
                d[2] = (unsigned char)min(255, max(0, s2 >> PRECISION_BITS));
            }
        }
    };
    if (a.ks_h <= 6) horizontal(std::integral_constant<int, 6>());
    else if (fast_h) horizontal(std::integral_constant<int, 12>());
    for (int it = threadIdx.x; !fast_h && it < ny * R; it += 256) {
        const int yy = it / R, ox = it - yy * R;
        const int xmin = bh[2 * ox], cnt = bh[2 * ox + 1];
        const uint8_t *row = src + ((long)(ymin + yy) * a.in_w + xmin) * 3;
        const int32_t *k = kh + ox * a.ks_h;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int t = 0; t < cnt; t++) {
            const int c = k[t];
            s0 += (int)row[3 * t] * c;
            s1 += (int)row[3 * t + 1] * c;
            s2 += (int)row[3 * t + 2] * c;
        }
        unsigned char *d = tmp + (yy * R + ox) * 3;
        d[0] = (unsigned char)min(255, max(0, s0 >> PRECISION_BITS));
        d[1] = (unsigned char)min(255, max(0, s1 >> PRECISION_BITS));
        d[2] = (unsigned char)min(255, max(0, s2 >> PRECISION_BITS));
    }
    __syncthreads();

    // ---- vertical pass, normalise, store: a thread takes two neighbouring pixels (their six bytes in one LDS
    // read per tap, one coefficient read for both; in the patch layout the pair shares a patch row, so every
    // value leaves as one 4-byte store of two 16-bit elements) ----
    const int p = a.patch, pp = p * p;
    const bool pairs = (R & 1) == 0 && (a.mode != EC_PRE_PATCHES16 || ((p & 1) == 0 && (a.kpad & 1) == 0));
    const int RW = pairs ? R / 2 : R;                  // work items per output row
    // Item order: in the patch layout the 14 x 14 values of one patch and channel are contiguous in the output row
    // (kpad elements per patch), so consecutive items walk (row in patch, pixel pair) INSIDE a patch and a wave's
    // 4-byte stores fall into 256 contiguous bytes; row-major over the band they fell into runs of 28 bytes, one per
    // patch, ten cache lines per store instruction.
    const bool by_patch = pairs && a.mode == EC_PRE_PATCHES16 && (oy1 - oy0) % p == 0;
    const int per_patch = p * (p / 2);
    for (int it = threadIdx.x; it < (oy1 - oy0) * RW; it += 256) {
        int oyl, ox;
        if (by_patch) {
            const int q = it / per_patch, r = it - q * per_patch;
            const int pyl = q / a.grid_w, pxl = q - pyl * a.grid_w;
            const int i = r / (p / 2), jp = r - i * (p / 2);
            oyl = pyl * p + i, ox = pxl * p + 2 * jp;
        } else {
            oyl = it / RW, ox = (it - oyl * RW) * (pairs ? 2 : 1);
        }
        const int oy = oy0 + oyl;
        const int y0 = bv[2 * oy] - ymin, cnt = bv[2 * oy + 1];
        const int32_t *k = kvs + oyl * a.ks_v;
        int sa[3], sb[3];
#pragma unroll
        for (int c = 0; c < 3; c++) sa[c] = sb[c] = 1 << (PRECISION_BITS - 1);
        const unsigned char *col = tmp + (y0 * R + ox) * 3;
        if (pairs) {
            for (int t = 0; t < cnt; t++) {
                const int c = k[t];
                unsigned long long v;
                __builtin_memcpy(&v, col, 8);                 // two pixels' six bytes (+ two beyond, inside the carve)
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    sa[ch] += (int)((v >> (8 * ch)) & 255u) * c;
                    sb[ch] += (int)((v >> (8 * (3 + ch))) & 255u) * c;
                }
                col += R * 3;
            }
        } else {
            for (int t = 0; t < cnt; t++) {
                const int c = k[t];
                unsigned v;
                __builtin_memcpy(&v, col, 4);
#pragma unroll
                for (int ch = 0; ch < 3; ch++) sa[ch] += (int)((v >> (8 * ch)) & 255u) * c;
                col += R * 3;
            }
        }
        int ua[3], ub[3];
        float va[3], vb[3];
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            ua[ch] = min(255, max(0, sa[ch] >> PRECISION_BITS));
            ub[ch] = min(255, max(0, sb[ch] >> PRECISION_BITS));
            va[ch] = lut[256 * ch + ua[ch]];
            vb[ch] = lut[256 * ch + ub[ch]];
        }
        if (a.mode == EC_PRE_CHW_F32) {
            float *o = (float *)a.out + (long)f * 3 * R * R + (long)oy * R + ox;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                if (pairs) *reinterpret_cast<float2 *>(o + (long)ch * R * R) = make_float2(va[ch], vb[ch]);
                else o[(long)ch * R * R] = va[ch];
            }
        } else if (a.mode == EC_PRE_HWC_U8) {
            uint8_t *o = (uint8_t *)a.out + ((long)f * R * R + (long)oy * R + ox) * 3;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                o[ch] = (uint8_t)ua[ch];
                if (pairs) o[3 + ch] = (uint8_t)ub[ch];
            }
        } else {
            const int py = oy / p, i = oy - py * p, px = ox / p, j = ox - px * p;
            // a patch row carries every value as hi + lo 16-bit parts: [hi (c,i,j) | lo (c,i,j) | 0]
            elem *o = (elem *)a.out + ((long)f * a.grid_w * a.grid_w + py * a.grid_w + px) * a.kpad +
                      i * p + j;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const elem ha = to16(va[ch], elem()), la = to16(va[ch] - (float)ha, elem());
                if (pairs) {
                    typedef elem elem2 __attribute__((ext_vector_type(2)));
                    const elem hb = to16(vb[ch], elem()), lb = to16(vb[ch] - (float)hb, elem());
                    elem2 hi, lo;
                    hi[0] = ha, hi[1] = hb, lo[0] = la, lo[1] = lb;
                    *reinterpret_cast<elem2 *>(o + ch * pp) = hi;
                    *reinterpret_cast<elem2 *>(o + (3 + ch) * pp) = lo;
                } else {
                    o[ch * pp] = ha;
                    o[(3 + ch) * pp] = la;
                }
            }
        }
    }
    if (a.mode == EC_PRE_PATCHES16) {
        // zero the K padding of the patches this band owns
        const int pad = a.kpad - 6 * pp;
        const int py0 = oy0 / p, npatch = ((oy1 - oy0) / p) * a.grid_w;
        for (int it = threadIdx.x; it < npatch * pad; it += 256) {
            const int q = it / pad, e = it - q * pad;
            elem *o = (elem *)a.out + ((long)f * a.grid_w * a.grid_w + py0 * a.grid_w + q) * a.kpad +
                      6 * pp + e;
            *o = to16(0.f, elem());
        }
    }
}

// fp32 [N, 3, R, R] -> 16-bit im2col rows [N, G, kpad]: [hi (c, i, j) | lo (c, i, j) | 0]
template <int DT>
__global__ __launch_bounds__(256) void patchify_kernel(const float *img, void *out, int n_img, int R,
                                                       int p, int kpad)
{
    typedef typename T16<DT>::elem elem;
    const int g = R / p, pp = p * p;
    const long total = (long)n_img * g * g * kpad;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
        const int e = (int)(it % kpad);
        const long q = it / kpad;
        const int px = (int)(q % g), py = (int)((q / g) % g);
        const long n = q / ((long)g * g);
        elem o = to16(0.f, elem());
        if (e < 6 * pp) {
            const int e3 = e < 3 * pp ? e : e - 3 * pp;
            const int c = e3 / pp, r = e3 - c * pp, i = r / p, j = r - i * p;
            const float v = img[((n * 3 + c) * R + (py * p + i)) * R + px * p + j];
            const elem hi = to16(v, elem());
            o = e < 3 * pp ? hi : to16(v - (float)hi, elem());
        }
        ((elem *)out)[it] = o;
    }
}

}  // namespace

extern "C" {

EC_API size_t ec_preprocess_plan_bytes(int in_h, int in_w, int n_px)
{
    if (in_h <= 0 || in_w <= 0 || n_px <= 0) return 0;
    int nh, nw;
    resized_size(in_h, in_w, n_px, &nh, &nw);
    const int ksh = ksize_for(in_w, nw), ksv = ksize_for(in_h, nh);
    const size_t words = P_HEADER + (size_t)n_px * 2 + (size_t)n_px * ksh + (size_t)n_px * 2 +
                         (size_t)n_px * ksv + 3 * 256;
    return words * 4;
}

EC_API int ec_preprocess_plan(int in_h, int in_w, int n_px, void *host_plan, size_t cap)
{
    EC_REQUIRE(in_h > 0 && in_w > 0 && n_px > 0, "ec_preprocess_plan: bad geometry");
    const size_t need = ec_preprocess_plan_bytes(in_h, in_w, n_px);
    EC_REQUIRE(host_plan && cap >= need, "ec_preprocess_plan: buffer %zu < %zu bytes", cap, need);
    int nh, nw;
    resized_size(in_h, in_w, n_px, &nh, &nw);
    // CenterCrop pads when the resized image is smaller than the crop; with
    // Resize(n_px) in front that cannot happen.
    EC_REQUIRE(nh >= n_px && nw >= n_px, "ec_preprocess_plan: resized %dx%d < crop %d", nh, nw, n_px);
    const int top = round_half_even_div2(nh - n_px), left = round_half_even_div2(nw - n_px);
    const int ksh = ksize_for(in_w, nw), ksv = ksize_for(in_h, nh);
    int32_t *pl = (int32_t *)host_plan;
    memset(pl, 0, need);
    pl[P_MAGIC] = PLAN_MAGIC, pl[P_IN_H] = in_h, pl[P_IN_W] = in_w, pl[P_NPX] = n_px;
    pl[P_NEW_H] = nh, pl[P_NEW_W] = nw, pl[P_TOP] = top, pl[P_LEFT] = left;
    pl[P_KS_H] = ksh, pl[P_KS_V] = ksv;
    int off = P_HEADER;
    pl[P_OFF_BH] = off, off += n_px * 2;
    pl[P_OFF_KH] = off, off += n_px * ksh;
    pl[P_OFF_BV] = off, off += n_px * 2;
    pl[P_OFF_KV] = off, off += n_px * ksv;
    pl[P_OFF_LUT] = off, off += 3 * 256;
    pl[P_WORDS] = off;
    coeffs(in_w, nw, left, n_px, ksh, pl + pl[P_OFF_BH], pl + pl[P_OFF_KH]);
    coeffs(in_h, nh, top, n_px, ksv, pl + pl[P_OFF_BV], pl + pl[P_OFF_KV]);
    // ToTensor (/255) then Normalize((x - mean) / std) in float32, torch's order
    const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f};
    const float stdv[3] = {0.26862954f, 0.26130258f, 0.27577711f};
    float *lut = reinterpret_cast<float *>(pl + pl[P_OFF_LUT]);
    for (int c = 0; c < 3; c++)
        for (int v = 0; v < 256; v++) {
            volatile float x = (float)v / 255.0f;
            volatile float y = x - mean[c];
            lut[c * 256 + v] = y / stdv[c];
        }
    return EC_OK;
}

EC_API int ec_preprocess(const uint8_t *frames, int F, const void *plan_host, const void *plan_dev,
                         void *out, int mode, int patch, int kpad, int dtype, ec_stream_t stream)
{
    EC_REQUIRE(F >= 0, "ec_preprocess: F=%d", F);
    if (F == 0) return EC_OK;
    EC_REQUIRE(frames && plan_host && plan_dev && out, "ec_preprocess: null buffer");
    const int32_t *pl = (const int32_t *)plan_host;
    EC_REQUIRE(pl[P_MAGIC] == PLAN_MAGIC, "ec_preprocess: not a plan");
    const int R = pl[P_NPX];
    PreArgs a;
    a.frames = frames, a.plan = (const int32_t *)plan_dev, a.out = out;
    a.in_h = pl[P_IN_H], a.in_w = pl[P_IN_W], a.n_px = R;
    a.ks_h = pl[P_KS_H], a.ks_v = pl[P_KS_V];
    a.off_bh = pl[P_OFF_BH], a.off_kh = pl[P_OFF_KH], a.off_bv = pl[P_OFF_BV];
    a.off_kv = pl[P_OFF_KV], a.off_lut = pl[P_OFF_LUT];
    a.mode = mode, a.patch = 1, a.kpad = 0, a.grid_w = 0;
    if (mode == EC_PRE_PATCHES16) {
        EC_REQUIRE(patch > 0 && R % patch == 0, "ec_preprocess: n_px %d not a multiple of patch %d", R,
                   patch);
        EC_REQUIRE(kpad >= 6 * patch * patch, "ec_preprocess: kpad %d < %d", kpad, 6 * patch * patch);
        a.patch = patch, a.kpad = kpad, a.grid_w = R / patch;
        a.band_rows = patch * (patch >= 32 ? 1 : (32 / patch));
    } else {
        EC_REQUIRE(mode == EC_PRE_CHW_F32 || mode == EC_PRE_HWC_U8, "ec_preprocess: mode %d", mode);
        a.band_rows = 32;
    }
    a.bands = ec::ceil_div(R, a.band_rows);
    // LDS: the tallest stack of horizontally resampled rows any band needs
    const int32_t *bv = pl + pl[P_OFF_BV];
    int max_ny = 0;
    for (int b = 0; b < a.bands; b++) {
        const int o0 = b * a.band_rows, o1 = (o0 + a.band_rows < R ? o0 + a.band_rows : R) - 1;
        const int ny = bv[2 * o1] + bv[2 * o1 + 1] - bv[2 * o0];
        if (ny > max_ny) max_ny = ny;
    }
    // ... then the normalisation table (3 x 256 floats) and the band's vertical coefficients
    a.tmp_bytes = ((max_ny * R * 3 + 8 + 15) / 16) * 16;      // (+ 8: the vertical pass reads 6 bytes as a qword)
    const int lds = a.tmp_bytes + 3 * 256 * 4 + ((a.band_rows * a.ks_v * 4 + 15) / 16) * 16;
    EC_REQUIRE(lds <= 160 * 1024, "ec_preprocess: band needs %d bytes of LDS", lds);
    hipStream_t s = static_cast<hipStream_t>(stream);
    auto kern = dtype == EC_BF16 ? preprocess_kernel<EC_BF16> : preprocess_kernel<EC_F16>;
    EC_REQUIRE(dtype == EC_F16 || dtype == EC_BF16, "ec_preprocess: unknown dtype %d", dtype);
    if (lds > 64 * 1024)
        EC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    // algorithmic bytes: uint8 frame in, n_px^2 x 3 values out
    const double out_bytes = mode == EC_PRE_CHW_F32 ? 12.0 * R * R
                             : mode == EC_PRE_HWC_U8 ? 3.0 * R * R
                                                     : 2.0 * (R / a.patch) * (R / a.patch) * a.kpad;
    ec::ProfScope prof(ec::PROF_PREPROCESS, s, 0, F * (3.0 * a.in_h * a.in_w + out_bytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)F * a.bands), dim3(256), lds, s, a);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

EC_API int ec_patchify(const float *img, int n_img, int n_px, int patch, int kpad, void *out16,
                       int dtype, ec_stream_t stream)
{
    EC_REQUIRE(n_img >= 0 && patch > 0 && n_px % patch == 0 && kpad >= 6 * patch * patch,
               "ec_patchify: bad geometry");
    if (n_img == 0) return EC_OK;
    EC_REQUIRE(img && out16, "ec_patchify: null buffer");
    const long total = (long)n_img * (n_px / patch) * (n_px / patch) * kpad;
    const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipStream_t s = static_cast<hipStream_t>(stream);
    ec::ProfScope prof(ec::PROF_PATCHIFY, s, 0, 12.0 * n_img * n_px * n_px + 2.0 * total);
    if (dtype == EC_F16)
        hipLaunchKernelGGL(patchify_kernel<EC_F16>, dim3(grid), dim3(256), 0, s, img, out16, n_img,
                           n_px, patch, kpad);
    else if (dtype == EC_BF16)
        hipLaunchKernelGGL(patchify_kernel<EC_BF16>, dim3(grid), dim3(256), 0, s, img, out16, n_img,
                           n_px, patch, kpad);
    else
        return ec::fail(EC_ERR_INVALID, "ec_patchify: unknown dtype %d", dtype);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

}  // extern "C"
