// events -> polarity-histogram frames on gfx950.
//
// Replaces /root/reference/datasets/vis.py make_event_histogram (:6-41) +
// parse_events (:44-52) as driven by events2frames (:75-117).
//
// One 1024-thread workgroup owns one frame.  The [rows, W, 2] uint32 histogram
// of a row band lives in LDS and is filled with LDS atomics; a frame that does
// not fit is walked band by band.  The reference needs two frame-wide
// reductions before it can colour a pixel (the hot-pixel threshold from
// mean/std, vis.py:17-24, then the max of what is left, vis.py:27), so a
// multi-band frame re-bins its events for each of the three passes instead of
// spilling a histogram to HBM.  To keep that off the memory system the events of
// the frame are first reduced to 4-byte bin indices in an LDS cache (20 000
// events = 80 KB beside a 36-row band for N-Caltech): HBM sees each event once,
// the re-binning passes scan LDS, and the only HBM write is the uint8 frame.
// Frames too long for the cache (N-ImageNet's 70 000 events) re-read their
// events from L2 / Infinity Cache; frames that fit one band (N-Cars) bin once.
//
// Exactness: counts are integers.  mean/std come from exact integer sums
// (numpy's float64 pairwise summation agrees to ~1e-15 relative; bins whose
// count sits within 1e-9 of the threshold are reported in stats.ambiguous).
// The float stage is the float64 sequence numpy >= 2 executes for
// vis.py:27-39, including the fused multiply-add of the dgemm behind
// `hist @ cmap`, with FP contraction disabled everywhere else.
#include <cstdlib>

#include "common.h"

// numpy evaluates every ufunc with one rounding per operation: no FMA contraction anywhere in
// this file (the one fused multiply-add of the reference is written out explicitly).
#pragma clang fp contract(off)

namespace {

constexpr int EV_THREADS = 1024;
constexpr int EV_WAVES = EV_THREADS / 64;
constexpr int EV_BIN_BYTES = 156 * 1024;          // histogram band
constexpr int EV_LUT_N = 16;                       // colour look-up table over counts 0..15 x 0..15
constexpr int EV_REDUCE_BYTES = 8 * EV_WAVES * 4;  // block-reduction scratch (u64 per wave, up to 3 values at once)
constexpr int EV_SCRATCH_BYTES = EV_REDUCE_BYTES + EV_LUT_N * EV_LUT_N * 4;   // + the LUT

struct EvArgs {
    const void *events;      // float4 (x, y, t, p) or packed 8-byte events
    const long long *range;  // [F,2]
    int H, W;
    double thresh;
    int count_non_zero, background_mask;
    double red[3], blue[3];
    uint8_t *frames;
    int *raw;
    int *kept;
    ec_frame_stats *stats;
    int rows_per_band, bands;
    int flip_x, negate_p; // test-time-augmentation views (utils.py:18-35)
    int float32_stage;    // vis.py:27-39 in float32 (numpy 1.x value-based casting) instead of float64
    int bin_bytes;       // LDS bytes of the histogram band (scratch and the event cache follow)
    int cache_events;    // capacity of the LDS event cache, 0 = none
    int F;               // frames; a workgroup takes frames blockIdx.x, + gridDim.x, ...
    unsigned *sort_ws;   // band-sorted bin indices, sort_cap per workgroup (long frames), or null
    int sort_cap;
    unsigned band_magic; // ceil(2^32 / rows_per_band): y / rows_per_band == (y * magic) >> 32 for y < 2^16
    unsigned *b10_ws;    // events_band10_kernel: [workgroup][band][wave][b10_cap] band-local bin codes
    int b10_rows, b10_bands, b10_cap, b10_events;   // rows per band, bands, region capacity, most events per frame
    uint8_t *redo;       // per-frame flags shared with events_pack10_kernel: that kernel sets redo[f] to 1 for
                         // a frame it could not finish (0 otherwise), this one then processes ONLY those frames
};

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned t = __shfl_down(v, o, 64);
        v = t > v ? t : v;
    }
    return v;
}

// 32-bit sums / maxima over a wave without LDS traffic: four DPP steps leave every lane with its row's (16 lanes)
// value, the four rows meet on the scalar unit.  (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror)
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v)
{
    v += dpp_u32<0xB1>(v), v += dpp_u32<0x4E>(v), v += dpp_u32<0x141>(v), v += dpp_u32<0x140>(v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 0) + (unsigned)__builtin_amdgcn_readlane((int)v, 16) +
           (unsigned)__builtin_amdgcn_readlane((int)v, 32) + (unsigned)__builtin_amdgcn_readlane((int)v, 48);
}
__device__ __forceinline__ unsigned wave_max_dpp(unsigned v)
{
    auto mx = [](unsigned x, unsigned y) { return x > y ? x : y; };
    v = mx(v, dpp_u32<0xB1>(v)), v = mx(v, dpp_u32<0x4E>(v)), v = mx(v, dpp_u32<0x141>(v)), v = mx(v, dpp_u32<0x140>(v));
    return mx(mx((unsigned)__builtin_amdgcn_readlane((int)v, 0), (unsigned)__builtin_amdgcn_readlane((int)v, 16)),
              mx((unsigned)__builtin_amdgcn_readlane((int)v, 32), (unsigned)__builtin_amdgcn_readlane((int)v, 48)));
}

// Sum over the workgroup, result in every thread.  scratch: EV_WAVES u64.
__device__ unsigned long long block_sum_u64(unsigned long long v, unsigned long long *scratch)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_sum_u64(v);
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    unsigned long long t = 0;
#pragma unroll
    for (int w = 0; w < EV_WAVES; w++) t += scratch[w];
    return t;
}

__device__ unsigned block_max_u32(unsigned v, unsigned long long *scratch)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_max_u32(v);
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    unsigned t = 0;
#pragma unroll
    for (int w = 0; w < EV_WAVES; w++) t = scratch[w] > t ? (unsigned)scratch[w] : t;
    return t;
}

// parse_events (vis.py:50: astype(int32) truncates toward zero), after the optional TTA
// transforms of utils.py: x -> W - 1 - x in float32 (:22), p -> -p (:34).
__device__ __forceinline__ void parse(const float4 e, int W, int flip_x, int negate_p, int &x, int &y,
                                      int &p)
{
    const float xf = flip_x ? (float)(W - 1) - e.x : e.x;
    x = (int)xf, y = (int)e.y, p = (int)(negate_p ? -e.w : e.w);
}

// Packed 8-byte event (include/eventclip_hip.h, EC_PACKED_*): x and y are already the truncated
// integers of parse_events, the polarity code is 0 (p == 0), 1 (p > 0) or 2 (p < 0).
typedef unsigned long long packed_t;
__device__ __forceinline__ void parse(const packed_t e, int W, int flip_x, int negate_p, int &x, int &y,
                                      int &p)
{
    x = (int)(e & 0xffffu), y = (int)((e >> 16) & 0xffffu);
    if (flip_x) x = W - 1 - x;
    const int code = (int)((e >> 32) & 3u);
    p = code == 0 ? 0 : ((code == 1) != (negate_p != 0) ? 1 : -1);
}

// vis.py:10-14 restricted to rows [y0, y1): LDS atomics, one per in-band event.
template <typename EV>
__device__ __forceinline__ void bin_one(const EV e, int y0, int y1, int H, int W, int flip_x,
                                        int negate_p, unsigned *bins, unsigned &dropped)
{
    int x, y, p;
    parse(e, W, flip_x, negate_p, x, y, p);
    if (p == 0) return;  // counted in neither channel (vis.py:10,12)
    if ((unsigned)x >= (unsigned)W || (unsigned)y >= (unsigned)H) {
        dropped++;
        return;
    }
    if (y >= y0 && y < y1) atomicAdd(&bins[((y - y0) * W + x) * 2 + (p < 0 ? 1 : 0)], 1u);
}

template <typename EV>
__device__ void bin_band(const EV *ev, long long n, int y0, int y1, int H, int W, int flip_x,
                         int negate_p, unsigned *bins, unsigned &dropped)
{
    const int nb = (y1 - y0) * W * 2;
    for (int i = threadIdx.x; i < nb; i += EV_THREADS) bins[i] = 0;
    __syncthreads();
    long long i = threadIdx.x;
    for (; i + 3 * EV_THREADS < n; i += 4 * EV_THREADS) {
        const EV e0 = ev[i], e1 = ev[i + EV_THREADS], e2 = ev[i + 2 * EV_THREADS],
                 e3 = ev[i + 3 * EV_THREADS];
        bin_one(e0, y0, y1, H, W, flip_x, negate_p, bins, dropped);
        bin_one(e1, y0, y1, H, W, flip_x, negate_p, bins, dropped);
        bin_one(e2, y0, y1, H, W, flip_x, negate_p, bins, dropped);
        bin_one(e3, y0, y1, H, W, flip_x, negate_p, bins, dropped);
    }
    for (; i < n; i += EV_THREADS) bin_one(ev[i], y0, y1, H, W, flip_x, negate_p, bins, dropped);
    __syncthreads();
}

// Event cache: every event of the frame is read from HBM ONCE, reduced to its bin index
// (((y * W + x) << 1) | channel, or EV_SKIP for p == 0 / outside the sensor) and kept in LDS,
// so the band / pass loops below re-scan 4 B per event from LDS instead of 16 B from L2.
constexpr unsigned EV_SKIP = 0xFFFFFFFFu;

template <typename EV>
__device__ void fill_cache(const EV *ev, long long n, int H, int W, int flip_x, int negate_p,
                           unsigned *cache, unsigned &dropped)
{
    auto encode = [&](const EV e) {
        int x, y, p;
        parse(e, W, flip_x, negate_p, x, y, p);
        if (p == 0) return EV_SKIP;
        if ((unsigned)x >= (unsigned)W || (unsigned)y >= (unsigned)H) {
            dropped++;
            return EV_SKIP;
        }
        return ((unsigned)(y * W + x) << 1) | (p < 0 ? 1u : 0u);
    };
    // eight loads in flight per thread: this phase is the frame's only HBM read and nothing overlaps it
    long long i = threadIdx.x;
    for (; i + 7 * EV_THREADS < n; i += 8 * EV_THREADS) {
        EV e[8];
#pragma unroll
        for (int k = 0; k < 8; k++) e[k] = ev[i + k * EV_THREADS];
#pragma unroll
        for (int k = 0; k < 8; k++) cache[i + k * EV_THREADS] = encode(e[k]);
    }
    for (; i < n; i += EV_THREADS) cache[i] = encode(ev[i]);
}

__device__ void bin_band_cached(const unsigned *cache, int n, int y0, int y1, int W, unsigned *bins)
{
    const int nb = (y1 - y0) * W * 2;
    const unsigned lo = (unsigned)(y0 * W * 2);
    for (int i = threadIdx.x; i < nb; i += EV_THREADS) bins[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += EV_THREADS) {
        const unsigned c = cache[i] - lo;          // EV_SKIP and other bands wrap far above nb
        if (c < (unsigned)nb) atomicAdd(&bins[c], 1u);
    }
    __syncthreads();
}

// Long frames (more events than the LDS cache holds, several bands: N-ImageNet's 70 000 events
// on 480 x 640) used to re-scan every event from L2 for each of the 3 passes x 16 bands.  Instead the
// events are bucketed ONCE by band into a global scratch slot owned by the workgroup (4-B bin
// indices; two scans of the events: count, then place), and each (pass, band) reads only its own
// contiguous bucket.  Slots inside a bucket come from 16 sub-counters per band (one per wave), so
// the LDS atomics that hand them out are spread like the binning atomics themselves.
constexpr int EV_SORT_MAX_BANDS = 64;
constexpr int EV_CC_N = 1024;      // count-of-counts histogram (sorted mode): how many bins hold the count c < 1023
constexpr int EV_SORT_BYTES = (EV_SORT_MAX_BANDS * EV_WAVES + EV_SORT_MAX_BANDS + 1 + EV_CC_N) * 4;

template <typename EV>
__device__ void sort_by_band(const EV *ev, long long n, int H, int W, int flip_x, int negate_p,
                             int bands, unsigned magic, unsigned *cnt, unsigned *start, unsigned *ws,
                             unsigned &dropped)
{
    const int wave = threadIdx.x >> 6;
    const int entries = bands * EV_WAVES;
    for (int i = threadIdx.x; i < entries; i += EV_THREADS) cnt[i] = 0;
    __syncthreads();
    // (both scans: eight loads in flight per thread -- one dependent load per iteration made each scan of a
    // 70 000-event frame 68 memory latencies long)
    unsigned out_of_sensor = 0;
    for (long long i = threadIdx.x; i < n; i += 8 * EV_THREADS) {
        EV e[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const long long j = i + (long long)k * EV_THREADS;
            e[k] = ev[j < n ? j : n - 1];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            int x, y, p;
            parse(e[k], W, flip_x, negate_p, x, y, p);
            const bool there = i + (long long)k * EV_THREADS < n && p != 0;
            const bool inside = (unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H;
            out_of_sensor += there && !inside;
            if (there && inside) {
                const unsigned band = (unsigned)(((unsigned long long)(unsigned)y * magic) >> 32);
                atomicAdd(&cnt[band * EV_WAVES + wave], 1u);
            }
        }
    }
    dropped += out_of_sensor;
    __syncthreads();
    // exclusive scan of the (band, wave) counts by the first wave: lane l owns a run of entries
    if (threadIdx.x < 64) {
        const int per = (entries + 63) / 64;
        const int lo = threadIdx.x * per, hi = min(entries, lo + per);
        unsigned sum = 0;
        for (int i = lo; i < hi; i++) sum += cnt[i];
        unsigned incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(incl, o, 64);
            if ((int)threadIdx.x >= o) incl += t;
        }
        unsigned run = incl - sum;
        for (int i = lo; i < hi; i++) {
            const unsigned c = cnt[i];
            cnt[i] = run;
            if ((i % EV_WAVES) == 0) start[i / EV_WAVES] = run;
            run += c;
        }
        if (threadIdx.x == 63) start[bands] = incl;
    }
    __syncthreads();
    for (long long i = threadIdx.x; i < n; i += 8 * EV_THREADS) {
        EV e[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const long long j = i + (long long)k * EV_THREADS;
            e[k] = ev[j < n ? j : n - 1];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            int x, y, p;
            parse(e[k], W, flip_x, negate_p, x, y, p);
            if (i + (long long)k * EV_THREADS >= n || p == 0 || (unsigned)x >= (unsigned)W || (unsigned)y >= (unsigned)H) continue;
            const unsigned band = (unsigned)(((unsigned long long)(unsigned)y * magic) >> 32);
            const unsigned slot = atomicAdd(&cnt[band * EV_WAVES + wave], 1u);
            ws[slot] = ((unsigned)(y * W + x) << 1) | (p < 0 ? 1u : 0u);
        }
    }
    __syncthreads();   // the stores are visible to the whole workgroup (one CU, one L1)
}

__device__ void bin_band_sorted(const unsigned *ws, unsigned begin, unsigned end, int y0, int y1, int W,
                                unsigned *bins)
{
    const int nb = (y1 - y0) * W * 2;
    const unsigned lo = (unsigned)(y0 * W * 2);
    for (int i = threadIdx.x; i < nb; i += EV_THREADS) bins[i] = 0;
    __syncthreads();
    for (unsigned i = begin + threadIdx.x; i < end; i += 4 * EV_THREADS) {      // four loads in flight
        unsigned c[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned j = i + k * EV_THREADS;
            c[k] = ws[j < end ? j : end - 1];
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (i + k * EV_THREADS < end) atomicAdd(&bins[c[k] - lo], 1u);
    }
    __syncthreads();
}

// Walks over a band's bucket of bin codes (four loads in flight), for the passes of long, sparse frames that
// touch only the bins the events name instead of zeroing and scanning the whole band: OP 0 counts the events into
// bins that are all zero, OP 1 reads every event's final count h back (the sum over EVENTS of h is the sum over
// BINS of h^2; cc[h] grows by h per bin with that count), OP 2 puts the touched bins back to zero.
template <int OP>
__device__ __forceinline__ void walk_bucket(const unsigned *ws, unsigned begin, unsigned end, unsigned lo, unsigned *bins,
                                            unsigned *cc, unsigned long long &s2)
{
    for (unsigned i = begin + threadIdx.x; i < end; i += 4 * EV_THREADS) {
        unsigned c[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned j = i + k * EV_THREADS;
            c[k] = ws[j < end ? j : end - 1];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (i + k * EV_THREADS >= end) continue;
            unsigned *bin = &bins[c[k] - lo];
            if (OP == 0) {
                atomicAdd(bin, 1u);
            } else if (OP == 1) {
                const unsigned h = *bin;
                s2 += h;
                atomicAdd(&cc[h < EV_CC_N - 1 ? h : EV_CC_N - 1], 1u);
            } else {
                *bin = 0;
            }
        }
    }
}

// vis.py:27-39 for one pixel in float32: what the reference's pinned numpy 1.25 computes (value-based
// casting keeps `float32 array / int64 scalar` and everything after it in float32; sgemm's K = 2
// inner product is fmaf(q, blue, p * red), every ufunc after it rounds once).  FP contraction is off.
__device__ __forceinline__ void colour_pixel_f32(unsigned c0, unsigned c1, float fmx, const EvArgs &a,
                                                 uint8_t out[3])
{
    const float p = (float)c0 / fmx;
    const float q = (float)c1 / fmx;
    float w = 0.f;
    if (a.background_mask) {
        w = p + q;
        if (w < 0.f) w = 0.f;
        if (w > 1.f) w = 1.f;
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float t = p * (float)a.red[c];
        float v = __builtin_fmaf(q, (float)a.blue[c], t);
        if (a.background_mask) {
            const float t1 = v * w;
            const float om = 1.f - w;
            const float t2 = 255.f * om;
            v = t1 + t2;
        }
        const float r = __builtin_rintf(v);
        out[c] = (r != r) ? (uint8_t)0 : (uint8_t)(int)r;
    }
}

// vis.py:27-39 for one pixel, float64, numpy's operation order.
__device__ __forceinline__ void colour_pixel(unsigned c0, unsigned c1, double dmx, const EvArgs &a,
                                             uint8_t out[3])
{
    if (a.float32_stage) return colour_pixel_f32(c0, c1, (float)dmx, a, out);
    const double p = (double)(float)c0 / dmx;  // vis.py:27 (astype(float32) then / int64 max)
    const double q = (double)(float)c1 / dmx;
    double w = 0.;
    if (a.background_mask) {                   // vis.py:35
        w = p + q;
        if (w < 0.) w = 0.;
        if (w > 1.) w = 1.;
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const double t = p * a.red[c];
        double v = __builtin_fma(q, a.blue[c], t);  // vis.py:31 (dgemm inner product)
        if (a.background_mask) {                     // vis.py:36-37
            const double t1 = v * w;
            const double om = 1. - w;
            const double t2 = 255. * om;
            v = t1 + t2;
        }
        const double r = __builtin_rint(v);          // vis.py:39, half to even
        out[c] = (r != r) ? (uint8_t)0 : (uint8_t)(int)r;
    }
}

// ---- hot-pixel threshold, vis.py:17-24 ----
// mean = S1/cnt; sum of squared deviations = S2 - S1^2/cnt = (cnt*S2 - S1^2)/cnt, exact in 128-bit
// integers, then numpy's own sequence: /cnt, sqrt, thresh*std + mean.
// The comparison `count > thr` (vis.py:24) in integers: for a non-negative integer h and a finite
// thr >= 0, h > thr <=> h > floor(thr); a NaN or infinite thr removes nothing.  The one count whose
// comparison could depend on numpy's summation order (|h - thr| <= 1e-9 |thr|) is rint(thr).
struct HotPixel {
    double thr;
    bool use_thr;
    unsigned thr_hi;     // counts above this are removed
    long long amb_h;     // the ambiguous count, or -1
};

__device__ __forceinline__ HotPixel hot_pixel_threshold(const EvArgs &a, unsigned long long s1,
                                                        unsigned long long s2, unsigned nnz, long long M2)
{
    HotPixel r;
    r.thr = __builtin_inf();
    r.use_thr = a.thresh > 0.;
    r.thr_hi = 0xFFFFFFFFu;
    r.amb_h = -1;
    bool spread = false;  // population not constant: numpy's summation order can matter
    if (r.use_thr) {
        const unsigned long long cnt = a.count_non_zero ? (unsigned long long)nnz : (unsigned long long)M2;
        if (cnt == 0) {
            r.thr = __builtin_nan("");  // empty population: comparisons are all false
        } else {
            const unsigned __int128 num = (unsigned __int128)cnt * s2 - (unsigned __int128)s1 * s1;
            spread = num != 0;
            const double mean = (double)s1 / (double)cnt;
            const double ss = (double)num / (double)cnt;
            const double var = ss / (double)cnt;
            const double sd = __builtin_sqrt(var);
            const double tsd = a.thresh * sd;
            r.thr = tsd + mean;
        }
    }
    if (r.use_thr && r.thr == r.thr && r.thr < 4294967295.) {
        r.thr_hi = (unsigned)__builtin_floor(r.thr);
        const double hr = __builtin_rint(r.thr);
        if (spread && __builtin_fabs(hr - r.thr) <= 1e-9 * __builtin_fabs(r.thr)) r.amb_h = (long long)hr;
    }
    return r;
}

template <typename EV>
__global__ __launch_bounds__(EV_THREADS) void events_to_frames_kernel(const EvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned *bins = reinterpret_cast<unsigned *>(smem);
    unsigned long long *scratch = reinterpret_cast<unsigned long long *>(smem + a.bin_bytes);
    unsigned *cache = reinterpret_cast<unsigned *>(smem + a.bin_bytes + EV_SCRATCH_BYTES);
    unsigned *sort_cnt = cache;                                   // sorted mode: no event cache
    unsigned *sort_start = cache + EV_SORT_MAX_BANDS * EV_WAVES;
    unsigned *cc = sort_start + EV_SORT_MAX_BANDS + 1;            // sorted mode: EV_CC_N words
    unsigned *ws = a.sort_ws ? a.sort_ws + (size_t)blockIdx.x * a.sort_cap : nullptr;

    if (a.redo) {       // behind events_pack10_kernel: ordinarily none of this workgroup's frames is flagged -- one look, together
        int any = 0;
        for (int f = blockIdx.x + (int)threadIdx.x * (int)gridDim.x; f < a.F; f += EV_THREADS * (int)gridDim.x) any |= a.redo[f];
        // (through the dynamic LDS the kernel owns: __syncthreads_or takes static LDS on top of the CU's 160 KB)
        const unsigned long long flagged = __ballot(any != 0);        // (every lane's frames: taken outside the branch)
        if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = flagged;
        __syncthreads();
        unsigned long long all = 0;
#pragma unroll
        for (int w = 0; w < EV_WAVES; w++) all |= scratch[w];
        if (all == 0) return;
        __syncthreads();
    }
  for (int f = blockIdx.x; f < a.F; f += gridDim.x) {
    if (a.redo && !a.redo[f]) continue;   // finished by events_pack10_kernel (workgroup-uniform)
    const long long e0 = a.range[2 * f], e1 = a.range[2 * f + 1];
    const EV *ev = reinterpret_cast<const EV *>(a.events) + e0;
    const long long n = e1 - e0;
    const int H = a.H, W = a.W;
    const long long M2 = (long long)H * W * 2;
    const int rpb = a.rows_per_band, bands = a.bands;

    // frames that fit the LDS event cache read their events from HBM exactly once
    const bool cached = n <= (long long)a.cache_events;
    // longer multi-band frames are bucketed by band into this workgroup's scratch slot
    const bool sorted = !cached && ws != nullptr && bands > 1 && n <= (long long)a.sort_cap;
    unsigned long long s1 = 0, s2 = 0;
    unsigned nnz = 0, dropped = 0;
    // Long frames re-bin every band for every pass; the second pass only wants the largest count that survives
    // the hot-pixel threshold (and how many bins hold the one ambiguous count).  Both follow from how many bins
    // hold each count, which pass 1 can tally on the side (non-zero bins only: ~ one LDS atomic per event): the
    // frame then takes 2 x bands re-binnings instead of 3 x.  Counts of 1023 and more share the last slot; a
    // frame that has one runs the real pass 2.
    const bool use_cc = sorted && !a.kept;
    // ... and a long frame is sparse (70 000 events on 614 400 bins): with the bins all zero at the start of a
    // band, pass 1 counts the bucket in, reads each event's count back and puts the touched bins back to zero --
    // three walks of ~4 events per thread instead of zeroing and scanning 38 000 words; pass 3 likewise clears what
    // it binned after colouring the band.  Not with the debug outputs, and not for a frame with a count >= 1023
    // (its tallies by count cannot be divided back into bins): those take the dense passes.
    bool lean = use_cc && !a.raw;
    if (use_cc) {
        for (int i = threadIdx.x; i < EV_CC_N; i += EV_THREADS) cc[i] = 0;   // (visible after the first band's barriers)
    }
    if (lean) {
        const int nbw = rpb * W * 2;
        for (int i = threadIdx.x; i < nbw; i += EV_THREADS) bins[i] = 0;
    }
    if (cached) {
        fill_cache(ev, n, H, W, a.flip_x, a.negate_p, cache, dropped);
        __syncthreads();
    } else if (sorted) {
        sort_by_band(ev, n, H, W, a.flip_x, a.negate_p, bands, a.band_magic, sort_cnt, sort_start, ws,
                     dropped);
    }
    if (lean) {
        for (int b = 0; b < bands; b++) {
            const unsigned lo = (unsigned)(b * rpb * W * 2), begin = sort_start[b], end = sort_start[b + 1];
            walk_bucket<0>(ws, begin, end, lo, bins, cc, s2);
            __syncthreads();
            walk_bucket<1>(ws, begin, end, lo, bins, cc, s2);
            __syncthreads();
            walk_bucket<2>(ws, begin, end, lo, bins, cc, s2);
            __syncthreads();
        }
        if (cc[EV_CC_N - 1] != 0) {          // (uniform) a count of 1023 or more: start over with the dense passes
            lean = false;
            s2 = 0;
            __syncthreads();
            for (int i = threadIdx.x; i < EV_CC_N; i += EV_THREADS) cc[i] = 0;
            __syncthreads();
        } else {
            // cc[h] holds h per bin with count h: back to bins per count, whose sum is the non-zero bins
            __syncthreads();
            for (int h = threadIdx.x; h < EV_CC_N - 1; h += EV_THREADS)
                if (h > 0) {
                    const unsigned per = cc[h] / (unsigned)h;
                    cc[h] = per;
                    nnz += per;
                }
            if (threadIdx.x == 0) s1 = sort_start[bands];      // every event that was binned
        }
    }
    // ---- pass 1: counts -> sum, sum of squares, non-zero bins ----
    for (int b = 0; b < bands && !lean; b++) {
        const int y0 = b * rpb, y1 = min(H, y0 + rpb);
        unsigned dr = 0;
        if (cached)
            bin_band_cached(cache, (int)n, y0, y1, W, bins);
        else if (sorted)
            bin_band_sorted(ws, sort_start[b], sort_start[b + 1], y0, y1, W, bins);
        else
            bin_band(ev, n, y0, y1, H, W, a.flip_x, a.negate_p, bins, dr);
        if (b == 0 && !cached && !sorted) dropped = dr;
        const int nb = (y1 - y0) * W * 2;
        for (int i = threadIdx.x; i < nb; i += EV_THREADS) {
            const unsigned h = bins[i];
            s1 += h;
            s2 += (unsigned long long)h * h;
            nnz += h > 0;
            if (use_cc && h > 0) atomicAdd(&cc[h < EV_CC_N - 1 ? h : EV_CC_N - 1], 1u);
            if (a.raw) a.raw[f * M2 + (long long)y0 * W * 2 + i] = (int)h;
        }
    }
    s1 = block_sum_u64(s1, scratch);
    s2 = block_sum_u64(s2, scratch);
    nnz = (unsigned)block_sum_u64(nnz, scratch);
    dropped = (unsigned)block_sum_u64(dropped, scratch);

    const HotPixel hp = hot_pixel_threshold(a, s1, s2, nnz, M2);
    const double thr = hp.thr;
    const bool use_thr = hp.use_thr;
    const unsigned thr_hi = hp.thr_hi;
    const long long amb_h = hp.amb_h;

    // ---- pass 2: max of the counts that survive (vis.py:24,27) ----
    unsigned mx = 0, amb = 0;
    const bool from_cc = use_cc && cc[EV_CC_N - 1] == 0;      // (workgroup-uniform: every thread reads the same word)
    if (from_cc) {
        for (int c = threadIdx.x; c < EV_CC_N - 1; c += EV_THREADS)
            if (c > 0 && cc[c] > 0 && (unsigned)c <= thr_hi) mx = (unsigned)c > mx ? (unsigned)c : mx;
        if (threadIdx.x == 0 && amb_h >= 0) {
            if (amb_h == 0) amb = (unsigned)(M2 - (long long)nnz);
            else if (amb_h < EV_CC_N - 1) amb = cc[amb_h];
        }
    }
    for (int b = 0; b < bands && !from_cc; b++) {
        const int y0 = b * rpb, y1 = min(H, y0 + rpb);
        unsigned dr = 0;
        if (bands > 1) {
            if (cached)
                bin_band_cached(cache, (int)n, y0, y1, W, bins);
            else if (sorted)
                bin_band_sorted(ws, sort_start[b], sort_start[b + 1], y0, y1, W, bins);
            else
                bin_band(ev, n, y0, y1, H, W, a.flip_x, a.negate_p, bins, dr);
        }
        const int nb = (y1 - y0) * W * 2;
        for (int i = threadIdx.x; i < nb; i += EV_THREADS) {
            unsigned h = bins[i];
            amb += (long long)h == amb_h;
            if (h > thr_hi) h = 0;
            mx = h > mx ? h : mx;
            if (bands == 1) bins[i] = h;  // single band: keep the thresholded counts for pass 3
            if (a.kept) a.kept[f * M2 + (long long)y0 * W * 2 + i] = (int)h;
        }
    }
    mx = block_max_u32(mx, scratch);
    amb = (unsigned)block_sum_u64(amb, scratch);
    const double dmx = (double)mx;

    if (a.stats && threadIdx.x == 0) {
        ec_frame_stats st;
        st.sum = s1;
        st.sumsq = s2;
        st.nnz = nnz;
        st.max_kept = mx;
        st.dropped = dropped;
        st.ambiguous = amb;
        st.thr = use_thr ? thr : __builtin_nan("");
        a.stats[f] = st;
    }

    // Event frames are mostly 0 .. few counts per pixel: the float64 colour stage is evaluated once per
    // frame for every pair of counts below 16 and looked up (same function, same bytes)
    unsigned *lut = reinterpret_cast<unsigned *>(smem + a.bin_bytes + EV_REDUCE_BYTES);
    if (threadIdx.x < EV_LUT_N * EV_LUT_N) {
        uint8_t px[4] = {0, 0, 0, 0};
        colour_pixel(threadIdx.x / EV_LUT_N, threadIdx.x % EV_LUT_N, dmx, a, px);
        lut[threadIdx.x] = (unsigned)px[0] | ((unsigned)px[1] << 8) | ((unsigned)px[2] << 16);
    }
    __syncthreads();
    auto colour = [&](unsigned h0, unsigned h1, uint8_t *px) {
        if (h0 > thr_hi) h0 = 0;                       // bands > 1: the counts were re-binned, threshold again
        if (h1 > thr_hi) h1 = 0;
        if (h0 < EV_LUT_N && h1 < EV_LUT_N) {
            const unsigned v = lut[h0 * EV_LUT_N + h1];
            px[0] = (uint8_t)v, px[1] = (uint8_t)(v >> 8), px[2] = (uint8_t)(v >> 16);
        } else {
            colour_pixel(h0, h1, dmx, a, px);
        }
    };

    // ---- pass 3: normalise, colour, blend, round -> uint8 (vis.py:27-39) ----
    uint8_t *out = a.frames + (long long)f * H * W * 3;
    for (int b = 0; b < bands; b++) {
        const int y0 = b * rpb, y1 = min(H, y0 + rpb);
        unsigned dr = 0;
        if (bands > 1) {
            if (cached)
                bin_band_cached(cache, (int)n, y0, y1, W, bins);
            else if (sorted && lean)
                walk_bucket<0>(ws, sort_start[b], sort_start[b + 1], (unsigned)(y0 * W * 2), bins, cc, s2);   // (bins all zero)
            else if (sorted)
                bin_band_sorted(ws, sort_start[b], sort_start[b + 1], y0, y1, W, bins);
            else
                bin_band(ev, n, y0, y1, H, W, a.flip_x, a.negate_p, bins, dr);
        }
        __syncthreads();
        const int npix = (y1 - y0) * W;
        uint8_t *o = out + (long long)y0 * W * 3;
        if ((W & 3) == 0) {
            // 4 pixels = 12 bytes = three dwords per thread, 4-byte aligned
            for (int g = threadIdx.x; g < npix / 4; g += EV_THREADS) {
                unsigned words[3];
                uint8_t *bytes = reinterpret_cast<uint8_t *>(words);
                const uint4 ca = *reinterpret_cast<const uint4 *>(&bins[g * 8]);
                const uint4 cb = *reinterpret_cast<const uint4 *>(&bins[g * 8 + 4]);
                unsigned c[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    colour(c[2 * k], c[2 * k + 1], bytes + 3 * k);
                }
                unsigned *dst = reinterpret_cast<unsigned *>(o + (long long)g * 12);
                dst[0] = words[0];
                dst[1] = words[1];
                dst[2] = words[2];
            }
        } else {
            for (int q = threadIdx.x; q < npix; q += EV_THREADS) {
                uint8_t px[3];
                colour(bins[2 * q], bins[2 * q + 1], px);
                o[3 * q] = px[0];
                o[3 * q + 1] = px[1];
                o[3 * q + 2] = px[2];
            }
        }
        __syncthreads();
        if (lean) {                                   // leave the band's bins zero for the next one
            walk_bucket<2>(ws, sort_start[b], sort_start[b + 1], (unsigned)(y0 * W * 2), bins, cc, s2);
            __syncthreads();
        }
    }
  }   // frames of this workgroup
}

// ---------------------------------------------------------------------------------------------
// Whole frame in LDS: three 10-bit counts per 32-bit word.
//
// A sensor of up to ~59 000 pixels (N-Caltech: 180 x 240 x 2 bins -> 113 KiB) holds its complete
// histogram on chip in this form: the events are read from HBM ONCE and binned ONCE with LDS atomics
// (no event cache, no bands).  The atomics RETURN the word as it was, and the statistics the reference
// takes from the finished histogram (sum, sum of squares, non-zero bins, max) are tallied from those
// previous counts event by event, so no pass walks the histogram for them; a pass for the max of what
// survives the hot-pixel threshold runs only in frames that lose a pixel to it, and the colour pass
// reads 12 pixels = 8 words at a time through a 256-entry table of the float stage.
// An event that finds its 10-bit field at 1023 overflows it (a pixel with more than 1023 events of one
// polarity): such a frame writes no pixels and sets redo[f] (every frame writes its flag, 0 or 1);
// events_to_frames_kernel, launched behind this kernel over the same frames, processes exactly those
// with 32-bit bins.
// Its own kernel rather than a mode of the one above so that its registers are its own (as a mode it
// pushed both paths into scratch spills).
// ---------------------------------------------------------------------------------------------
constexpr unsigned P10_MASK = 1023u;

#ifdef EC_EVENTS_DIAG
// phase stamps of the first 4096 frames of a launch (diagnostic build only; tools/events_phases.py)
__device__ unsigned long long g_ev_phase[4096 * 8];
#define EV_STAMP(k)                                                                   \
    do {                                                                              \
        if (threadIdx.x == 0 && f >= 0 && f < 4096) g_ev_phase[f * 8 + (k)] = wall_clock64(); \
    } while (0)
#else
#define EV_STAMP(k)
#endif

// the frame's statistics (four sums and a max of per-thread tallies) behind one pair of barriers
struct P10Stats {
    unsigned long long total, s2, nnz, dropped;
    unsigned max;
};
__device__ __forceinline__ P10Stats block_stats(unsigned hmax, unsigned total, unsigned s2, unsigned nnz, unsigned dropped,
                                                unsigned long long *scratch)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // per-wave sums of the 32-bit partials fit 32 bits (s2: <= 64 threads x 20 events x 2047)
    const unsigned long long p01 = (unsigned long long)wave_sum_u32(total) | ((unsigned long long)wave_sum_u32(nnz) << 32);
    const unsigned long long p2 = wave_sum_u32(s2);
    const unsigned long long p3 = (unsigned long long)wave_sum_u32(dropped) | ((unsigned long long)wave_max_dpp(hmax) << 32);
    __syncthreads();
    if (lane == 0) scratch[wave] = p01, scratch[EV_WAVES + wave] = p2, scratch[2 * EV_WAVES + wave] = p3;
    __syncthreads();
    P10Stats r = {0, 0, 0, 0, 0};
#pragma unroll
    for (int w = 0; w < EV_WAVES; w++) {
        const unsigned long long a01 = scratch[w], a3 = scratch[2 * EV_WAVES + w];
        r.total += a01 & 0xffffffffull, r.nnz += a01 >> 32, r.s2 += scratch[EV_WAVES + w];
        r.dropped += a3 & 0xffffffffull;
        const unsigned m = (unsigned)(a3 >> 32);
        r.max = m > r.max ? m : r.max;
    }
    return r;
}

template <typename EV>
__global__ __launch_bounds__(EV_THREADS) void events_pack10_kernel(const EvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned *bins = reinterpret_cast<unsigned *>(smem);
    uint4 *bins4 = reinterpret_cast<uint4 *>(smem);
    unsigned long long *scratch = reinterpret_cast<unsigned long long *>(smem + a.bin_bytes);
    unsigned *lut = reinterpret_cast<unsigned *>(smem + a.bin_bytes + EV_REDUCE_BYTES);
    const int H = a.H, W = a.W;
    const int M2 = H * W * 2;
    const int words = (M2 + 2) / 3;
    const int words4 = (words + 3) / 4;                  // (a.bin_bytes is a multiple of 16)
    constexpr int PF = 4;                                // events per thread per round; two rounds in flight
    constexpr int ROUND = PF * EV_THREADS;
    constexpr unsigned EV_NOT_BINNED = 0xFFFFFFFFu, EV_DROPPED = 0xFFFFFFFEu;

    // One workgroup per CU (the histogram takes the CU's LDS) walks frames blockIdx.x, + gridDim.x, ...  A frame's
    // events come in rounds of 4 096 through two register buffers: while one round is binned the next is in flight,
    // and behind a frame's last rounds the buffers are refilled with the first two rounds of the NEXT frame, which
    // land behind the statistics, the threshold and the LDS passes.  The workgroups of a launch drift apart on their
    // own, so the memory system sees reads all the time instead of in bursts.
    EV ea[PF], eb[PF];
    auto request = [&](EV (&e)[PF], const EV *ev, long long n, long long i) {   // clamped: a masked slot re-reads the last event
#pragma unroll
        for (int k = 0; k < PF; k++) {
            const long long j = i + (long long)k * EV_THREADS;
            e[k] = ev[j < n ? j : n - 1];
        }
    };
    // (the frame ranges through the constant address space: scalar loads.  As ordinary global memory -- the kernel
    // stores to other global memory -- they were vector loads behind an s_waitcnt vmcnt(0) at the top of every
    // frame, which also waited for the previous frame's pixel stores)
    typedef const __attribute__((address_space(4))) long long *range_ptr;
    const range_ptr range = (range_ptr)(a.range);
    // events of a frame that this kernel bins: a frame of more than 2^24 events is left to the 32-bit kernel (the
    // per-wave tallies are sized for less)
    auto binnable = [](long long n) { return n > (1ll << 24) ? 0ll : n; };
    // (the walk starts one frame early with an empty frame of no output, whose only effect is the request for the
    // first real frame's rounds: one place per buffer that loads events -- with more, the compiler kept copies of the
    // buffers alive and spilled)
    long long e0 = 0, n = 0;
    bool dirty = true;                                   // the histogram is not all zero
    for (int f = (int)blockIdx.x - (int)gridDim.x; f < a.F; f += gridDim.x) {
        const bool real = f >= 0;
        EV_STAMP(0);
        const EV *ev = reinterpret_cast<const EV *>(a.events) + e0;
        const int fn = f + (int)gridDim.x;               // the frame after this one
        long long e0n = 0, nn = 0;
        if (fn < a.F) e0n = range[2 * fn], nn = range[2 * fn + 1] - e0n;
        const EV *evn = reinterpret_cast<const EV *>(a.events) + e0n;

        if (dirty) {                                     // (ordinarily the colour pass leaves it zero, below)
            for (int i = threadIdx.x; i < words4; i += EV_THREADS) bins4[i] = make_uint4(0, 0, 0, 0);
            __syncthreads();
        }
        dirty = real;
        EV_STAMP(1);
        // The statistics the reference takes from the finished histogram (vis.py:17-24: sum, sum of squares,
        // non-zero bins; vis.py:27: max) come out of the binning itself: the atomic returns the word as it was, h =
        // the field's count before this event, and over the events of a frame  sum of (2 h + 1) = sum over bins of
        // count^2, number of h == 0 = non-zero bins, max of h + 1 = largest count -- no pass over the 28 800 words.
        // An event that finds h == 1023 overflows its 10-bit field: the first overflow of a word sees the true 1023
        // (fields are only ever wrong after one), so `some event saw 1023` == `some field overflowed`.
        // (the tallies are selects on the lambda's return value, not increments inside its divergent branches)
        unsigned dropped = 0, binned = 0, s2t = 0, nnzt = 0, hmax = 0;
        auto bin_event = [&](const EV ek) -> unsigned {     // previous count of the event's bin, or one of the codes
            int x, y, p;
            parse(ek, W, a.flip_x, a.negate_p, x, y, p);
            const bool inside = (unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H;
            unsigned h = p == 0 || inside ? EV_NOT_BINNED : EV_DROPPED;
            if (p != 0 && inside) {
                const unsigned bin = (unsigned)(y * W + x) * 2u + (p < 0 ? 1u : 0u);
                const unsigned w = bin / 3u, sh = 10u * (bin - 3u * w);
                h = (atomicAdd(&bins[w], 1u << sh) >> sh) & P10_MASK;
            }
            return h;
        };
        const bool too_long = n > (1ll << 24);
        const long long nb = binnable(n), nnb = binnable(nn);
        // Every thread runs the same number of rounds, two per iteration (slots past the frame's end are masked, a
        // round past it is skipped): the frame's only HBM read.
        const int pairs = nb > 0 ? (int)((nb + 2 * ROUND - 1) / (2 * ROUND)) : 1;
        auto bin_round = [&](const EV (&e)[PF], int r) {
            const long long i = threadIdx.x + (long long)r * ROUND;
            unsigned h[PF];
#pragma unroll
            for (int k = 0; k < PF; k++) h[k] = i + (long long)k * EV_THREADS < nb ? bin_event(e[k]) : EV_NOT_BINNED;
#pragma unroll
            for (int k = 0; k < PF; k++) {
                const bool there = h[k] < EV_DROPPED;
                binned += there ? 1u : 0u;
                dropped += h[k] == EV_DROPPED ? 1u : 0u;
                s2t += there ? 2u * h[k] + 1u : 0u;
                nnzt += h[k] == 0u ? 1u : 0u;
                const unsigned now = there ? h[k] + 1u : 0u;   // 1024 = this event overflowed its field
                hmax = now > hmax ? now : hmax;
            }
        };
        // round g of this frame into a buffer, or -- past this frame's iterations -- round g - 2 pairs of the next
        auto refill = [&](EV (&e)[PF], int g) {
            const bool mine = g < 2 * pairs;
            const long long first = (long long)(mine ? g : g - 2 * pairs) * ROUND;
            const EV *src = mine ? ev : evn;
            const long long ns = mine ? nb : nnb;
            if (first < ns) request(e, src, ns, first + threadIdx.x);
        };
        for (int t = 0; t < pairs; t++) {
            bin_round(ea, 2 * t);
            refill(ea, 2 * t + 2);
            bin_round(eb, 2 * t + 1);
            refill(eb, 2 * t + 3);
        }
        EV_STAMP(2);
        if (!real) {                                     // the lead-in frame: nothing but the requests above
            e0 = e0n, n = nn;
            continue;
        }
        const P10Stats sums = block_stats(hmax, binned, s2t, nnzt, dropped, scratch);   // (its first barrier ends the binning)
        const bool overflow = too_long || sums.max > P10_MASK;   // a field overflowed: the 32-bit kernel takes the frame
        if (threadIdx.x == 0) a.redo[f] = overflow ? 1 : 0;
        const unsigned gmax = sums.max;
        const unsigned long long s1 = sums.total, s2 = sums.s2;
        const unsigned nnz = (unsigned)sums.nnz;
        dropped = (unsigned)sums.dropped;
        const HotPixel hp = hot_pixel_threshold(a, s1, s2, nnz, M2);
        const unsigned thr_hi = hp.thr_hi;
        EV_STAMP(3);

        if (!overflow) {
        // ---- pass 2: max of the counts that survive; debug outputs ----
        // Only when the answer is not known already: the largest count survives the threshold in an ordinary frame
        // (then it is the max of what is left, vis.py:27), and no count sits within 1e-9 of the threshold.
        unsigned mx = gmax, amb = 0;
        const bool amb_possible = hp.amb_h >= 0 && hp.amb_h <= (long long)gmax;
        if (a.raw || a.kept || amb_possible) {               // debug outputs, or a count within 1e-9 of the threshold
            mx = 0;
            for (int w = threadIdx.x; w < words; w += EV_THREADS) {
                const unsigned v = bins[w];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const int idx = 3 * w + k;
                    unsigned h = (v >> (10 * k)) & P10_MASK;
                    if (a.raw && idx < M2) a.raw[(long long)f * M2 + idx] = (int)h;
                    amb += (long long)h == hp.amb_h;
                    if (h > thr_hi) h = 0;
                    mx = h > mx ? h : mx;
                    if (a.kept && idx < M2) a.kept[(long long)f * M2 + idx] = (int)h;
                }
            }
            // max and the ambiguous-count tally behind one pair of barriers
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            mx = wave_max_dpp(mx);
            amb = wave_sum_u32(amb);
            __syncthreads();
            if (lane == 0) scratch[wave] = (unsigned long long)mx | ((unsigned long long)amb << 32);
            __syncthreads();
            mx = 0, amb = 0;
#pragma unroll
            for (int w = 0; w < EV_WAVES; w++) {
                const unsigned long long t = scratch[w];
                const unsigned m = (unsigned)(t & 0xffffffffull);
                mx = m > mx ? m : mx;
                amb += (unsigned)(t >> 32);
            }
        } else if (gmax > thr_hi) {
            // hot pixels were removed: the largest count that is left.  t = h - (thr_hi + 1) wraps to the top of the
            // unsigned range exactly for the survivors, in their order, so one unsigned max per field finds it
            // (thr_hi < gmax <= 1023 here)
            const unsigned off = thr_hi + 1u;
            unsigned t = 0;
            for (int i = threadIdx.x; i < words4; i += EV_THREADS) {
                const uint4 q = bins4[i];
                const unsigned v[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const unsigned t0 = (v[j] & P10_MASK) - off, t1 = ((v[j] >> 10) & P10_MASK) - off,
                                   t2 = ((v[j] >> 20) & P10_MASK) - off;
                    const unsigned m01 = t0 > t1 ? t0 : t1, m2 = t2 > t ? t2 : t;
                    t = m01 > m2 ? m01 : m2;
                }
            }
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            t = wave_max_dpp(t);
            __syncthreads();
            if (lane == 0) scratch[wave] = t;
            __syncthreads();
            t = 0;
#pragma unroll
            for (int w = 0; w < EV_WAVES; w++) t = (unsigned)scratch[w] > t ? (unsigned)scratch[w] : t;
            mx = t >= 0x80000000u ? t + off : 0u;
        }
        const double dmx = (double)mx;
        if (a.stats && threadIdx.x == 0) {
            ec_frame_stats st;
            st.sum = s1;
            st.sumsq = s2;
            st.nnz = nnz;
            st.max_kept = mx;
            st.dropped = dropped;
            st.ambiguous = amb;
            st.thr = hp.use_thr ? hp.thr : __builtin_nan("");
            a.stats[f] = st;
        }

        // ---- pass 3: colour through the per-frame look-up table for small counts ----
        // lut[h0 | h1 << 4] = the pixel of the counts (h0, h1) < 16 AFTER the threshold (a count above it is 0), so
        // the look-up path does not compare against the threshold at all
        EV_STAMP(4);
        if (threadIdx.x < EV_LUT_N * EV_LUT_N) {
            uint8_t px[4] = {0, 0, 0, 0};
            const unsigned h0 = threadIdx.x % EV_LUT_N, h1 = threadIdx.x / EV_LUT_N;
            colour_pixel(h0 > thr_hi ? 0u : h0, h1 > thr_hi ? 0u : h1, dmx, a, px);
            lut[threadIdx.x] = (unsigned)px[0] | ((unsigned)px[1] << 8) | ((unsigned)px[2] << 16);
        }
        __syncthreads();
        EV_STAMP(5);
        uint8_t *out = a.frames + (long long)f * H * W * 3;
        const int npix = H * W;
        // one pixel the long way: counts cut out of the words wherever they lie, threshold, table or float stage,
        // three byte stores.  (One copy of the float stage in the loop: unrolled into the group loop below it took
        // the kernel past 128 registers, and a spill next to the prefetch waits for the prefetch.)
        auto pixel = [&](int q) {
            const unsigned b0 = 2u * q, wa = b0 / 3u, wb = (b0 + 1u) / 3u;
            unsigned h0 = (bins[wa] >> (10u * (b0 - 3u * wa))) & P10_MASK, h1 = (bins[wb] >> (10u * (b0 + 1u - 3u * wb))) & P10_MASK;
            unsigned v;
            if (h0 < EV_LUT_N && h1 < EV_LUT_N) {
                v = lut[h0 | (h1 << 4)];
            } else {
                if (h0 > thr_hi) h0 = 0;
                if (h1 > thr_hi) h1 = 0;
                uint8_t px[3];
                colour_pixel(h0, h1, dmx, a, px);
                v = (unsigned)px[0] | ((unsigned)px[1] << 8) | ((unsigned)px[2] << 16);
            }
            out[3 * q] = (uint8_t)v, out[3 * q + 1] = (uint8_t)(v >> 8), out[3 * q + 2] = (uint8_t)(v >> 16);
        };
        // 12 pixels = 24 counts = exactly 8 words (two 16-byte LDS reads, every field at a fixed place), 36 bytes =
        // nine dwords out; frames whose byte size is not a multiple of 4 take the byte path.  A group whose 24 counts
        // are all below 16 (bits 4..9 of every field clear) is twelve look-ups with the index cut straight out of
        // the words; any other group goes pixel by pixel.
        // A frame that is all groups leaves its histogram zero for the next frame: each group clears the words it read.
        const int groups = (((long long)npix * 3) & 3) == 0 ? npix / 12 : 0;
        const bool clears = groups * 12 == npix;
        dirty = !clears;
        for (int g = threadIdx.x; g < groups; g += EV_THREADS) {
            const uint4 qa = bins4[2 * g], qb = bins4[2 * g + 1];
            const unsigned w[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
            const unsigned any = (w[0] | w[1]) | (w[2] | w[3]) | (w[4] | w[5]) | (w[6] | w[7]);
            if ((any & 0x3F0FC3F0u) == 0) {
                unsigned c[12];
#pragma unroll
                for (int k = 0; k < 12; k++) {
                    const int fa = 2 * k, wa = fa / 3, pa = fa % 3;      // pixel k: fields 2 k and 2 k + 1
                    unsigned idx;
                    if (pa == 0) idx = (w[wa] & 0xFu) | ((w[wa] >> 6) & 0xF0u);
                    else if (pa == 1) idx = ((w[wa] >> 10) & 0xFu) | ((w[wa] >> 16) & 0xF0u);
                    else idx = ((w[wa] >> 20) & 0xFu) | ((w[wa + 1] & 0xFu) << 4);
                    c[k] = lut[idx];
                }
                unsigned *dst = reinterpret_cast<unsigned *>(out + (long long)g * 36);
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    dst[3 * k] = c[4 * k] | (c[4 * k + 1] << 24);
                    dst[3 * k + 1] = (c[4 * k + 1] >> 8) | (c[4 * k + 2] << 16);
                    dst[3 * k + 2] = (c[4 * k + 2] >> 16) | (c[4 * k + 3] << 8);
                }
            } else {
#pragma unroll 1
                for (int k = 0; k < 12; k++) pixel(12 * g + k);
            }
            if (clears) bins4[2 * g] = make_uint4(0, 0, 0, 0), bins4[2 * g + 1] = make_uint4(0, 0, 0, 0);
        }
        for (int q = groups * 12 + threadIdx.x; q < npix; q += EV_THREADS) pixel(q);
        }   // !overflow
        EV_STAMP(6);
        __syncthreads();                                 // the passes are done with the histogram and the table
        e0 = e0n, n = nn;
    }
}

// ---------------------------------------------------------------------------------------------
// Large sensors (N-ImageNet: 480 x 640, 70 000 events per frame): row bands of 10-bit counts.
//
// The frame's histogram (614 400 counts) does not fit a CU, so the frame is walked in row bands of packed 10-bit
// counts (81 rows = 135 KiB: 6 bands, where 32-bit bins take 17) and its events are routed to their bands ONCE, in the
// scan that reads them from HBM: each wave appends the band-local bin codes (4 B) of its events to a region of its own
// per band in the caller's workspace ([band][wave][cap] per workgroup -- sized for the worst case, every event in one
// band, so there is no counting scan and no prefix sum; the lanes of a wave that meet in a band find their slots
// from one ballot per band and one LDS add by the band's first lane).  Every later walk of a band is each wave
// reading its own region back, coalesced, from L2.
// As in the whole-frame kernel the statistics come from the binning (returning atomics); the max of what survives the
// hot-pixel threshold and the tally of the one ambiguous count follow from how many events found their bin at count
// v - 1 (= bins with a final count >= v), counted per value v on the side (v <= 4 with ballots, the rare rest with an
// LDS add), so there is no pass over the bands for them.  The colour pass re-bins a band (fire-and-forget atomics),
// colours 12 pixels = 8 words at a time and leaves the band zero.  An overflowing 10-bit field, or a frame with more
// events than the regions were sized for, flags the frame for events_to_frames_kernel.
// ---------------------------------------------------------------------------------------------
constexpr int B10_MAX_BANDS = 16;
constexpr int B10_CNT_BYTES = EV_WAVES * B10_MAX_BANDS * 4;
constexpr int B10_CC_BYTES = (EV_CC_N + 4) * 4;          // events per count value 0 .. 1024
constexpr int B10_SIDE_BYTES = EV_SCRATCH_BYTES + B10_CNT_BYTES + B10_CC_BYTES;

template <typename EV>
__global__ __launch_bounds__(EV_THREADS) void events_band10_kernel(const EvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned *bins = reinterpret_cast<unsigned *>(smem);
    uint4 *bins4 = reinterpret_cast<uint4 *>(smem);
    unsigned long long *scratch = reinterpret_cast<unsigned long long *>(smem + a.bin_bytes);
    unsigned *lut = reinterpret_cast<unsigned *>(smem + a.bin_bytes + EV_REDUCE_BYTES);
    unsigned *cnt = lut + EV_LUT_N * EV_LUT_N;           // [wave][band]: codes in the wave's region of the band
    unsigned *cc = cnt + EV_WAVES * B10_MAX_BANDS;       // [v]: events that brought their bin to the count v
    const int H = a.H, W = a.W;
    const long long M2 = (long long)H * W * 2;
    const int rpb = a.b10_rows, bands = a.b10_bands, cap = a.b10_cap;
    const int band_words4 = (int)(((long long)rpb * W * 2 + 2) / 3 + 3) / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned *region = a.b10_ws + ((size_t)blockIdx.x * bands * EV_WAVES + wave) * cap;   // + band * EV_WAVES * cap
    const size_t band_stride = (size_t)EV_WAVES * cap;
    constexpr int PF = 4;
    constexpr int ROUND = PF * EV_THREADS;
    constexpr unsigned NO_BAND = 0xFFu;

    EV ea[PF], eb[PF];                                   // (the two-buffer scheme of events_pack10_kernel)
    auto request = [&](EV (&e)[PF], const EV *ev, long long n, long long i) {
#pragma unroll
        for (int k = 0; k < PF; k++) {
            const long long j = i + (long long)k * EV_THREADS;
            e[k] = ev[j < n ? j : n - 1];
        }
    };
    typedef const __attribute__((address_space(4))) long long *range_ptr;
    const range_ptr range = (range_ptr)(a.range);
    const long long most = a.b10_events < (1 << 24) ? a.b10_events : (1 << 24);
    auto routable = [&](long long n) { return n > most ? 0ll : n; };
    long long e0 = 0, n = 0;
    bool dirty = true;
    for (int f = (int)blockIdx.x - (int)gridDim.x; f < a.F; f += gridDim.x) {
        const bool real = f >= 0;
        const EV *ev = reinterpret_cast<const EV *>(a.events) + e0;
        const int fn = f + (int)gridDim.x;
        long long e0n = 0, nn = 0;
        if (fn < a.F) e0n = range[2 * fn], nn = range[2 * fn + 1] - e0n;
        const EV *evn = reinterpret_cast<const EV *>(a.events) + e0n;
        const bool too_long = n > most;
        const long long nb = routable(n), nnb = routable(nn);

        if (dirty) {
            for (int i = threadIdx.x; i < band_words4; i += EV_THREADS) bins4[i] = make_uint4(0, 0, 0, 0);
            dirty = false;
        }
        for (int i = threadIdx.x; i < EV_CC_N + 4; i += EV_THREADS) cc[i] = 0;
        if (lane < B10_MAX_BANDS) cnt[wave * B10_MAX_BANDS + lane] = 0;       // (only this wave touches them)

        // ---- the scan: the frame's only HBM read; every event to its band's region of this wave ----
        unsigned dropped = 0;
        auto place = [&](const EV ek, bool there) {
            int x, y, p;
            parse(ek, W, a.flip_x, a.negate_p, x, y, p);
            const bool inside = (unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H;
            const bool ok = there && p != 0 && inside;
            dropped += there && p != 0 && !inside ? 1u : 0u;
            const unsigned band = ok ? (unsigned)(((unsigned long long)(unsigned)y * a.band_magic) >> 32) : NO_BAND;
            unsigned long long mine = 0;                 // the lanes whose event falls in this lane's band
            for (int b = 0; b < bands; b++) {
                const unsigned long long m = __ballot(band == (unsigned)b);
                if (band == (unsigned)b) mine = m;
            }
            const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mine >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mine, 0u));
            unsigned base = 0;
            if (ok && rank == 0) base = atomicAdd(&cnt[wave * B10_MAX_BANDS + band], (unsigned)__popcll(mine));
            base = __shfl(base, ok ? __ffsll((long long)mine) - 1 : lane, 64);
            if (ok) {
                const unsigned local = (unsigned)((y - (int)band * rpb) * W + x) * 2u + (p < 0 ? 1u : 0u);
                region[band * band_stride + base + rank] = local;
            }
        };
        const int pairs = nb > 0 ? (int)((nb + 2 * ROUND - 1) / (2 * ROUND)) : 1;
        auto place_round = [&](const EV (&e)[PF], int r) {
            const long long i = threadIdx.x + (long long)r * ROUND;
#pragma unroll
            for (int k = 0; k < PF; k++) place(e[k], i + (long long)k * EV_THREADS < nb);
        };
        auto refill = [&](EV (&e)[PF], int g) {
            const bool mine = g < 2 * pairs;
            const long long first = (long long)(mine ? g : g - 2 * pairs) * ROUND;
            const EV *src = mine ? ev : evn;
            const long long ns = mine ? nb : nnb;
            if (first < ns) request(e, src, ns, first + threadIdx.x);
        };
        for (int t = 0; t < pairs; t++) {
            place_round(ea, 2 * t);
            refill(ea, 2 * t + 2);
            place_round(eb, 2 * t + 1);
            refill(eb, 2 * t + 3);
        }
        if (!real) {                                     // the lead-in frame: nothing but the requests above
            e0 = e0n, n = nn;
            __syncthreads();
            continue;
        }
        __syncthreads();                                 // codes stored, bins / cc zero

        // ---- pass 1: every band binned once for the statistics ----
        unsigned binned = 0, s2t = 0, nnzt = 0, hmax = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
        for (int b = 0; b < bands; b++) {
            const unsigned count = cnt[wave * B10_MAX_BANDS + b];
            const unsigned *codes = region + b * band_stride;
            for (unsigned i0 = 0; i0 < count; i0 += 4 * 64) {          // four loads in flight; wave-uniform trips
                const unsigned i = i0 + lane;
                unsigned c[4], h[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const unsigned j = i + k * 64;
                    c[k] = codes[j < count ? j : count - 1];
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const bool there = i + k * 64 < count;
                    h[k] = 0xFFFFFFFFu;
                    if (there) {
                        const unsigned w = c[k] / 3u, sh = 10u * (c[k] - 3u * w);
                        h[k] = (atomicAdd(&bins[w], 1u << sh) >> sh) & P10_MASK;
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const bool there = h[k] != 0xFFFFFFFFu;
                    const unsigned now = there ? h[k] + 1u : 0u;       // 1024 = this event overflowed its field
                    binned += there ? 1u : 0u;
                    s2t += there ? 2u * h[k] + 1u : 0u;
                    nnzt += h[k] == 0u ? 1u : 0u;
                    hmax = now > hmax ? now : hmax;
                    // events per value of `now`: the common small values by ballot (a wave's lanes would all hit the
                    // same LDS word), the rest by an LDS add
                    c1 += (unsigned)__popcll(__ballot(now == 1u)), c2 += (unsigned)__popcll(__ballot(now == 2u));
                    c3 += (unsigned)__popcll(__ballot(now == 3u)), c4 += (unsigned)__popcll(__ballot(now == 4u));
                    if (now >= 5u) atomicAdd(&cc[now], 1u);
                }
            }
            __syncthreads();
            for (int i = threadIdx.x; i < band_words4; i += EV_THREADS) bins4[i] = make_uint4(0, 0, 0, 0);
            __syncthreads();
        }
        if (lane == 0) atomicAdd(&cc[1], c1), atomicAdd(&cc[2], c2), atomicAdd(&cc[3], c3), atomicAdd(&cc[4], c4);
        const P10Stats sums = block_stats(hmax, binned, s2t, nnzt, dropped, scratch);   // (its barriers publish cc)
        const bool overflow = too_long || sums.max > P10_MASK;
        if (threadIdx.x == 0) a.redo[f] = overflow ? 1 : 0;
        if (!overflow) {
            const unsigned gmax = sums.max;
            const unsigned long long s1 = sums.total, s2 = sums.s2;
            const unsigned nnz = (unsigned)sums.nnz;
            const HotPixel hp = hot_pixel_threshold(a, s1, s2, nnz, M2);
            const unsigned thr_hi = hp.thr_hi;
            // cc[v] events brought a bin to v = bins whose final count is >= v, so cc[v] - cc[v + 1] bins end at v
            unsigned mx = gmax, amb = 0;
            if (gmax > thr_hi) {
                const unsigned v = threadIdx.x;            // 1024 threads: one count value each
                unsigned cand = v >= 1u && v <= thr_hi && cc[v] > cc[v + 1] ? v : 0u;
                cand = wave_max_dpp(cand);
                __syncthreads();
                if (lane == 0) scratch[wave] = cand;
                __syncthreads();
                mx = 0;
#pragma unroll
                for (int w = 0; w < EV_WAVES; w++) mx = (unsigned)scratch[w] > mx ? (unsigned)scratch[w] : mx;
            }
            if (hp.amb_h == 0) amb = (unsigned)(M2 - (long long)nnz);
            else if (hp.amb_h > 0 && hp.amb_h <= (long long)P10_MASK) amb = cc[hp.amb_h] - cc[hp.amb_h + 1];
            const double dmx = (double)mx;
            if (a.stats && threadIdx.x == 0) {
                ec_frame_stats st;
                st.sum = s1;
                st.sumsq = s2;
                st.nnz = nnz;
                st.max_kept = mx;
                st.dropped = (unsigned)sums.dropped;
                st.ambiguous = amb;
                st.thr = hp.use_thr ? hp.thr : __builtin_nan("");
                a.stats[f] = st;
            }
            if (threadIdx.x < EV_LUT_N * EV_LUT_N) {     // the table of events_pack10_kernel: lut[h0 | h1 << 4], thresholded
                uint8_t px[4] = {0, 0, 0, 0};
                const unsigned h0 = threadIdx.x % EV_LUT_N, h1 = threadIdx.x / EV_LUT_N;
                colour_pixel(h0 > thr_hi ? 0u : h0, h1 > thr_hi ? 0u : h1, dmx, a, px);
                lut[threadIdx.x] = (unsigned)px[0] | ((unsigned)px[1] << 8) | ((unsigned)px[2] << 16);
            }
            // ---- pass 3: every band binned again and coloured ----
            const bool aligned = (((long long)H * W * 3) & 3) == 0;        // every frame starts on a dword
            for (int b = 0; b < bands; b++) {
                const unsigned count = cnt[wave * B10_MAX_BANDS + b];
                const unsigned *codes = region + b * band_stride;
                for (unsigned i0 = 0; i0 < count; i0 += 4 * 64) {
                    const unsigned i = i0 + lane;
                    unsigned c[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const unsigned j = i + k * 64;
                        c[k] = codes[j < count ? j : count - 1];
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (i + k * 64 < count) {
                            const unsigned w = c[k] / 3u, sh = 10u * (c[k] - 3u * w);
                            atomicAdd(&bins[w], 1u << sh);
                        }
                }
                __syncthreads();                         // (also: the table is written)
                const int y0 = b * rpb, rows = min(H - y0, rpb);
                const int npix = rows * W;
                uint8_t *out = a.frames + ((long long)f * H + y0) * W * 3;
                auto pixel = [&](int q) {                // (events_pack10_kernel's: one pixel the long way)
                    const unsigned b0 = 2u * q, wa = b0 / 3u, wb = (b0 + 1u) / 3u;
                    unsigned h0 = (bins[wa] >> (10u * (b0 - 3u * wa))) & P10_MASK, h1 = (bins[wb] >> (10u * (b0 + 1u - 3u * wb))) & P10_MASK;
                    unsigned v;
                    if (h0 < EV_LUT_N && h1 < EV_LUT_N) {
                        v = lut[h0 | (h1 << 4)];
                    } else {
                        if (h0 > thr_hi) h0 = 0;
                        if (h1 > thr_hi) h1 = 0;
                        uint8_t px[3];
                        colour_pixel(h0, h1, dmx, a, px);
                        v = (unsigned)px[0] | ((unsigned)px[1] << 8) | ((unsigned)px[2] << 16);
                    }
                    out[3 * q] = (uint8_t)v, out[3 * q + 1] = (uint8_t)(v >> 8), out[3 * q + 2] = (uint8_t)(v >> 16);
                };
                // (a band of rpb rows starts on a multiple of 36 bytes: rpb * W is a multiple of 12 by the host's plan)
                const int groups = aligned && (((long long)y0 * W * 3) & 3) == 0 ? npix / 12 : 0;
                const bool clears = groups * 12 == npix;
                for (int g = threadIdx.x; g < groups; g += EV_THREADS) {
                    const uint4 qa = bins4[2 * g], qb = bins4[2 * g + 1];
                    const unsigned w[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
                    const unsigned any = (w[0] | w[1]) | (w[2] | w[3]) | (w[4] | w[5]) | (w[6] | w[7]);
                    if (any == 0) {                      // a long frame is sparse: mostly this
                        const unsigned z = lut[0];
                        unsigned *dst = reinterpret_cast<unsigned *>(out + (long long)g * 36);
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            dst[3 * k] = z | (z << 24);
                            dst[3 * k + 1] = (z >> 8) | (z << 16);
                            dst[3 * k + 2] = (z >> 16) | (z << 8);
                        }
                        continue;
                    }
                    if ((any & 0x3F0FC3F0u) == 0) {
                        unsigned c[12];
#pragma unroll
                        for (int k = 0; k < 12; k++) {
                            const int fa = 2 * k, wa = fa / 3, pa = fa % 3;
                            unsigned idx;
                            if (pa == 0) idx = (w[wa] & 0xFu) | ((w[wa] >> 6) & 0xF0u);
                            else if (pa == 1) idx = ((w[wa] >> 10) & 0xFu) | ((w[wa] >> 16) & 0xF0u);
                            else idx = ((w[wa] >> 20) & 0xFu) | ((w[wa + 1] & 0xFu) << 4);
                            c[k] = lut[idx];
                        }
                        unsigned *dst = reinterpret_cast<unsigned *>(out + (long long)g * 36);
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            dst[3 * k] = c[4 * k] | (c[4 * k + 1] << 24);
                            dst[3 * k + 1] = (c[4 * k + 1] >> 8) | (c[4 * k + 2] << 16);
                            dst[3 * k + 2] = (c[4 * k + 2] >> 16) | (c[4 * k + 3] << 8);
                        }
                    } else {
#pragma unroll 1
                        for (int k = 0; k < 12; k++) pixel(12 * g + k);
                    }
                    if (clears) bins4[2 * g] = make_uint4(0, 0, 0, 0), bins4[2 * g + 1] = make_uint4(0, 0, 0, 0);
                }
                for (int q = groups * 12 + threadIdx.x; q < npix; q += EV_THREADS) pixel(q);
                __syncthreads();
                if (!clears) {
                    for (int i = threadIdx.x; i < band_words4; i += EV_THREADS) bins4[i] = make_uint4(0, 0, 0, 0);
                    __syncthreads();
                }
            }
        }
        __syncthreads();
        e0 = e0n, n = nn;
    }
}

// center_events, datasets/utils.py:38-57, one workgroup per sample, in place:
// t -= min t; x -= ((x_max + x_min + 1) - W) // 2; y likewise (float32 arithmetic).
// (1024 threads, four events in flight per thread per pass: one workgroup per sample has to keep a CU's share of
// the HBM bandwidth busy on its own)
__global__ __launch_bounds__(1024) void center_events_kernel(float4 *events, const long long *range,
                                                            int H, int W)
{
    __shared__ float red[5][16];
    const long long e0 = range[2 * blockIdx.x], e1 = range[2 * blockIdx.x + 1];
    float4 *ev = events + e0;
    const long long n = e1 - e0;
    float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY, tmin = INFINITY;
    long long i = threadIdx.x;
    for (; i + 3 * 1024 < n; i += 4 * 1024) {
        float4 e[4];
#pragma unroll
        for (int u = 0; u < 4; u++) e[u] = ev[i + u * 1024];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            xmin = fminf(xmin, e[u].x), xmax = fmaxf(xmax, e[u].x);
            ymin = fminf(ymin, e[u].y), ymax = fmaxf(ymax, e[u].y);
            tmin = fminf(tmin, e[u].z);
        }
    }
    for (; i < n; i += 1024) {
        const float4 e = ev[i];
        xmin = fminf(xmin, e.x), xmax = fmaxf(xmax, e.x);
        ymin = fminf(ymin, e.y), ymax = fmaxf(ymax, e.y);
        tmin = fminf(tmin, e.z);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        xmin = fminf(xmin, __shfl_xor(xmin, o, 64)), xmax = fmaxf(xmax, __shfl_xor(xmax, o, 64));
        ymin = fminf(ymin, __shfl_xor(ymin, o, 64)), ymax = fmaxf(ymax, __shfl_xor(ymax, o, 64));
        tmin = fminf(tmin, __shfl_xor(tmin, o, 64));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        red[0][wave] = xmin, red[1][wave] = xmax, red[2][wave] = ymin, red[3][wave] = ymax,
        red[4][wave] = tmin;
    __syncthreads();
    xmin = red[0][0], xmax = red[1][0], ymin = red[2][0], ymax = red[3][0], tmin = red[4][0];
#pragma unroll
    for (int w = 1; w < 16; w++) {
        xmin = fminf(xmin, red[0][w]), xmax = fmaxf(xmax, red[1][w]);
        ymin = fminf(ymin, red[2][w]), ymax = fmaxf(ymax, red[3][w]);
        tmin = fminf(tmin, red[4][w]);
    }
    const float xs = floorf(((xmax + xmin + 1.f) - (float)W) / 2.f);   // utils.py:53
    const float ys = floorf(((ymax + ymin + 1.f) - (float)H) / 2.f);   // utils.py:54
    i = threadIdx.x;
    for (; i + 3 * 1024 < n; i += 4 * 1024) {
        float4 e[4];
#pragma unroll
        for (int u = 0; u < 4; u++) e[u] = ev[i + u * 1024];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            e[u].x -= xs, e[u].y -= ys, e[u].z -= tmin;
            ev[i + u * 1024] = e[u];
        }
    }
    for (; i < n; i += 1024) {
        float4 e = ev[i];
        e.x -= xs, e.y -= ys, e.z -= tmin;
        ev[i] = e;
    }
}


// center_events on packed events: the same shift in integers (coordinates are integral, so
// utils.py:53-54's float floor-division equals the arithmetic shift of the integer sum);
// a coordinate shifted below zero wraps to >= 32768 and is dropped by the binning kernel like
// any other out-of-sensor event.  t is kept relative to the sample's first event.
__global__ __launch_bounds__(1024) void center_packed_kernel(packed_t *events, const long long *range,
                                                            int H, int W)
{
    __shared__ unsigned red[5][16];
    const long long e0 = range[2 * blockIdx.x], e1 = range[2 * blockIdx.x + 1];
    packed_t *ev = events + e0;
    const long long n = e1 - e0;
    unsigned xmin = ~0u, xmax = 0, ymin = ~0u, ymax = 0, tmin = ~0u;
    auto see = [&](packed_t e) {
        const unsigned x = (unsigned)(e & 0xffffu), y = (unsigned)((e >> 16) & 0xffffu), t = (unsigned)(e >> 34);
        xmin = min(xmin, x), xmax = max(xmax, x), ymin = min(ymin, y), ymax = max(ymax, y);
        tmin = min(tmin, t);
    };
    long long i = threadIdx.x;
    for (; i + 7 * 1024 < n; i += 8 * 1024) {
        packed_t e[8];
#pragma unroll
        for (int u = 0; u < 8; u++) e[u] = ev[i + u * 1024];
#pragma unroll
        for (int u = 0; u < 8; u++) see(e[u]);
    }
    for (; i < n; i += 1024) see(ev[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        xmin = min(xmin, (unsigned)__shfl_xor((int)xmin, o, 64));
        xmax = max(xmax, (unsigned)__shfl_xor((int)xmax, o, 64));
        ymin = min(ymin, (unsigned)__shfl_xor((int)ymin, o, 64));
        ymax = max(ymax, (unsigned)__shfl_xor((int)ymax, o, 64));
        tmin = min(tmin, (unsigned)__shfl_xor((int)tmin, o, 64));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        red[0][wave] = xmin, red[1][wave] = xmax, red[2][wave] = ymin, red[3][wave] = ymax,
        red[4][wave] = tmin;
    __syncthreads();
    xmin = red[0][0], xmax = red[1][0], ymin = red[2][0], ymax = red[3][0], tmin = red[4][0];
#pragma unroll
    for (int w = 1; w < 16; w++) {
        xmin = min(xmin, red[0][w]), xmax = max(xmax, red[1][w]);
        ymin = min(ymin, red[2][w]), ymax = max(ymax, red[3][w]);
        tmin = min(tmin, red[4][w]);
    }
    const int xs = ((int)(xmax + xmin + 1) - W) >> 1;   // floor division by 2 (utils.py:53)
    const int ys = ((int)(ymax + ymin + 1) - H) >> 1;   // utils.py:54
    auto moved = [&](packed_t e) {
        const unsigned x = ((unsigned)(e & 0xffffu) - (unsigned)xs) & 0xffffu;
        const unsigned y = ((unsigned)((e >> 16) & 0xffffu) - (unsigned)ys) & 0xffffu;
        const packed_t t = (e >> 34) - tmin;
        return (packed_t)x | ((packed_t)y << 16) | (e & (3ull << 32)) | (t << 34);
    };
    i = threadIdx.x;
    for (; i + 7 * 1024 < n; i += 8 * 1024) {
        packed_t e[8];
#pragma unroll
        for (int u = 0; u < 8; u++) e[u] = ev[i + u * 1024];
#pragma unroll
        for (int u = 0; u < 8; u++) ev[i + u * 1024] = moved(e[u]);
    }
    for (; i < n; i += 1024) ev[i] = moved(ev[i]);
}

// float32 (x, y, t, p) -> packed: parse_events' truncating casts (vis.py:50), t in microseconds
// rounded to nearest and saturated to 30 bits.  Events whose coordinates are not integral or do
// not fit 16 bits cannot be represented: they are counted in *bad and packed with polarity code 0
// (binned nowhere).
__global__ __launch_bounds__(256) void pack_events_kernel(const float4 *events, long long n,
                                                          packed_t *out, unsigned *bad)
{
    unsigned nbad = 0;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += 256ll * gridDim.x) {
        const float4 e = events[i];
        const int x = (int)e.x, y = (int)e.y, p = (int)e.w;
        unsigned code = p == 0 ? 0u : (p > 0 ? 1u : 2u);
        const bool ok = (float)x == e.x && (float)y == e.y && x >= 0 && y >= 0 && x < 65536 && y < 65536;
        if (!ok) nbad++, code = 0;
        double tu = __builtin_rint((double)e.z * 1e6);
        tu = tu < 0. ? 0. : (tu > 1073741823. ? 1073741823. : tu);
        out[i] = (packed_t)((unsigned)x & 0xffffu) | ((packed_t)((unsigned)y & 0xffffu) << 16) |
                 ((packed_t)code << 32) | ((packed_t)(unsigned)tu << 34);
    }
    if (bad) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nbad += __shfl_down(nbad, o, 64);
        if ((threadIdx.x & 63) == 0 && nbad) atomicAdd(bad, nbad);
    }
}

}  // namespace

extern "C" EC_API int ec_center_events(float *events, const int64_t *sample_range, int B, int H, int W,
                                       ec_stream_t stream)
{
    EC_REQUIRE(B >= 0 && H > 0 && W > 0, "ec_center_events: bad arguments");
    if (B == 0) return EC_OK;
    EC_REQUIRE(events && sample_range, "ec_center_events: null buffer");
    EC_REQUIRE(((uintptr_t)events & 15) == 0, "ec_center_events: events must be 16-byte aligned");
    hipLaunchKernelGGL(center_events_kernel, dim3(B), dim3(1024), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<float4 *>(events),
                       reinterpret_cast<const long long *>(sample_range), H, W);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

namespace {
// EC_EVENTS_NO_BAND10 / EC_EVENTS_NO_PACK10 (fall back to the 32-bit band kernel: A/B timing) are read by the diagnostic build
// only; the product library's kernel choice never depends on the process environment
#ifdef EC_EVENTS_DIAG
bool events_env_off(const char *name) { return getenv(name) != nullptr; }
#else
constexpr bool events_env_off(const char *) { return false; }
#endif


// LDS bytes of the whole-frame 10-bit histogram, and whether that path applies: the frame fits as packed
// counts next to the reduction scratch and the LUT.  (Also for sensors whose 32-bit histogram would fit:
// the packed kernel takes its statistics from the binning and keeps the next frame's events in flight,
// and two of its workgroups share a CU when the histogram is small, N-Cars: 32 KB.)
inline long pack10_bytes(int H, int W) { return (((long)H * W * 2 + 2) / 3 * 4 + 15) / 16 * 16; }
inline bool pack10_fits(int H, int W) { return pack10_bytes(H, W) + EV_SCRATCH_BYTES <= 160 * 1024; }

// Row bands of 10-bit counts for sensors whose whole histogram does not fit (events_band10_kernel): rows per band a
// multiple of 12 / gcd(W, 12) so that every band is whole groups of 12 pixels, as many as fit the LDS next to the
// side arrays, then evened out over the bands.
constexpr long B10_FLAG_BYTES = 65536;
struct B10Plan {
    bool ok;
    int rows, bands, cap;
    long bin_bytes;
    size_t region_bytes;     // per workgroup
};
inline B10Plan b10_plan(int H, int W, int max_frame_events)
{
    B10Plan p = {false, 0, 0, 0, 0, 0};
    if (max_frame_events <= 0 || pack10_fits(H, W)) return p;
    int gcd = W % 12, t = 12;
    while (gcd) { const int r = t % gcd; t = gcd; gcd = r; }          // t = gcd(W, 12)
    const int unit = 12 / t;
    const long budget_fields = ((long)160 * 1024 - B10_SIDE_BYTES) / 16 * 16 / 4 * 3;
    int rows = (int)(budget_fields / ((long)W * 2)) / unit * unit;
    if (rows < unit) return p;
    const int bands = ec::ceil_div(H, rows);
    if (bands > B10_MAX_BANDS) return p;
    rows = ec::ceil_div(ec::ceil_div(H, bands), unit) * unit;
    p.rows = rows, p.bands = ec::ceil_div(H, rows);
    p.bin_bytes = (((long)rows * W * 2 + 2) / 3 * 4 + 15) / 16 * 16;
    p.cap = 64 * ec::ceil_div(max_frame_events, EV_THREADS);
    p.region_bytes = (size_t)p.bands * EV_WAVES * p.cap * 4;
    p.ok = p.bin_bytes + B10_SIDE_BYTES <= (long)160 * 1024;
    return p;
}

template <typename EV>
int launch_events(const void *events, const int64_t *frame_range, int F, const ec_events_params *prm,
                  uint8_t *frames, int32_t *raw_counts, int32_t *kept_counts, ec_frame_stats *stats,
                  ec_stream_t stream)
{
    EC_REQUIRE(prm != nullptr, "ec_events_to_frames: params is null");
    EC_REQUIRE(F >= 0, "ec_events_to_frames: F=%d", F);
    if (F == 0) return EC_OK;
    EC_REQUIRE(events && frame_range && frames, "ec_events_to_frames: null buffer");
    EC_REQUIRE(prm->H > 0 && prm->W > 0, "ec_events_to_frames: bad shape (%d,%d)", prm->H, prm->W);
    EC_REQUIRE(((uintptr_t)events & (sizeof(EV) - 1)) == 0,
               "ec_events_to_frames: events must be %d-byte aligned", (int)sizeof(EV));
    const int row_bytes = prm->W * 2 * 4;
    EC_REQUIRE(row_bytes <= EV_BIN_BYTES, "ec_events_to_frames: W=%d too wide for one LDS row band",
               prm->W);

    EvArgs a;
    a.events = events;
    a.range = reinterpret_cast<const long long *>(frame_range);
    a.H = prm->H;
    a.W = prm->W;
    a.thresh = prm->thresh;
    a.count_non_zero = prm->count_non_zero;
    a.flip_x = prm->flip_x, a.negate_p = prm->negate_p;
    a.float32_stage = prm->float32_stage;
    a.background_mask = prm->background_mask;
    for (int c = 0; c < 3; c++) {
        a.red[c] = (double)(float)prm->red[c];    // cmap.astype(float32), vis.py:30
        a.blue[c] = (double)(float)prm->blue[c];
    }
    a.frames = frames;
    a.raw = raw_counts;
    a.kept = kept_counts;
    a.stats = stats;
    // LDS plan: [histogram band | reduction scratch | event cache or sort counters].  With the
    // per-frame event count bounded (max_frame_events, known to the caller from
    // split_event_count's N) the cache takes 4 B per event and the band gets what is left.  Frames
    // too long for that are bucketed by band in the caller's sort workspace (one slot of
    // max_frame_events indices per resident workgroup, which then walks several frames); without
    // a workspace, or for frames longer than the bound, every band pass re-reads the events from L2.
    constexpr int LDS_TOTAL = 160 * 1024;
    int cache_events = 0, bin_budget = EV_BIN_BYTES;
    if (prm->max_frame_events > 0) {
        const long cache_bytes = ((long)prm->max_frame_events * 4 + 15) / 16 * 16;
        const long left = (long)EV_BIN_BYTES - cache_bytes;
        // worth it when the whole frame then needs at most 8 bands
        if (left >= row_bytes && ec::ceil_div(prm->H, (int)(left / row_bytes)) <= 8) {
            cache_events = prm->max_frame_events;
            bin_budget = (int)left;
        }
    }
    int grid = F;
    a.sort_ws = nullptr;
    a.sort_cap = 0;
    // banded 10-bit path (events_band10_kernel): the workspace is [frame flags | code regions, one per CU]; what the
    // 32-bit kernel behind it uses as sort slots for the frames it is left with is the region area
    const B10Plan b10 = b10_plan(prm->H, prm->W, prm->max_frame_events);
    const int cus_now = ec::cu_count() > 0 ? ec::cu_count() : 256;
    const bool use_b10 = b10.ok && !raw_counts && !kept_counts && prm->sort_workspace &&
                         prm->sort_workspace_bytes >= (size_t)B10_FLAG_BYTES + b10.region_bytes && !events_env_off("EC_EVENTS_NO_BAND10");
    unsigned char *sort_base = static_cast<unsigned char *>(prm->sort_workspace) + (use_b10 ? B10_FLAG_BYTES : 0);
    const size_t sort_bytes = prm->sort_workspace ? prm->sort_workspace_bytes - (use_b10 ? B10_FLAG_BYTES : 0) : 0;
    if (cache_events == 0 && prm->max_frame_events > 0 && prm->sort_workspace) {
        const int sort_budget = (LDS_TOTAL - EV_SCRATCH_BYTES - EV_SORT_BYTES) / 16 * 16;
        const int budget = sort_budget < EV_BIN_BYTES ? sort_budget : EV_BIN_BYTES;
        const int rows = budget / row_bytes;
        const size_t slot = (size_t)prm->max_frame_events * 4;
        const size_t slots = sort_bytes / slot;
        if (rows >= 1 && ec::ceil_div(prm->H, rows) > 1 && ec::ceil_div(prm->H, rows) <= EV_SORT_MAX_BANDS &&
            slots >= 1) {
            EC_REQUIRE(((uintptr_t)prm->sort_workspace & 3) == 0, "ec_events_to_frames: sort workspace alignment");
            bin_budget = budget;
            a.sort_ws = reinterpret_cast<unsigned *>(sort_base);
            a.sort_cap = prm->max_frame_events;
            int cus = 256;
            int dev = 0;
            if (hipGetDevice(&dev) == hipSuccess)
                (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            grid = F < cus ? F : cus;
            if ((size_t)grid > slots) grid = (int)slots;
        }
    }
    const int max_rows = bin_budget / row_bytes;
    a.bands = ec::ceil_div(prm->H, max_rows);
    a.rows_per_band = ec::ceil_div(prm->H, a.bands);
    a.bin_bytes = (a.rows_per_band * row_bytes + 15) / 16 * 16;
    a.cache_events = cache_events;
    a.F = F;
    a.band_magic = (unsigned)((0x100000000ull + (unsigned)a.rows_per_band - 1) / (unsigned)a.rows_per_band);

    const int lds = LDS_TOTAL;   // always the full carve: one attribute call, one workgroup per CU
    if (int rc = ec::ensure_dynamic_lds(reinterpret_cast<const void *>(events_to_frames_kernel<EV>), lds))
        return rc;
    hipStream_t hs = static_cast<hipStream_t>(stream);
    // algorithmic bytes: 16 B (packed: 8 B) per event in + 3*H*W out.  The frame
    // ranges live on the device; the caller states their total length in prm->total_events (0 =
    // unknown: only the output part is reported then)
    ec::ProfScope prof(ec::PROF_EVENTS, hs, 0,
                       (double)F * prm->H * prm->W * 3.0 +
                           (double)(prm->total_events > 0 ? prm->total_events : 0) * sizeof(EV));
    a.redo = nullptr;
    if (use_b10) {
        // banded 10-bit path first, then the 32-bit kernel for the frames it flagged (none, for ordinary data)
        const int b10_lds = (int)b10.bin_bytes + B10_SIDE_BYTES;
        if (int rc = ec::ensure_dynamic_lds(reinterpret_cast<const void *>(events_band10_kernel<EV>), lds)) return rc;
        uint8_t *flags = static_cast<uint8_t *>(prm->sort_workspace);
        size_t wgs_fit = (prm->sort_workspace_bytes - B10_FLAG_BYTES) / b10.region_bytes;
        if (wgs_fit > (size_t)cus_now) wgs_fit = (size_t)cus_now;
        for (long f0 = 0; f0 < F; f0 += B10_FLAG_BYTES) {
            const int fc = (int)(F - f0 < B10_FLAG_BYTES ? F - f0 : B10_FLAG_BYTES);
            EvArgs g = a;
            g.range = a.range + 2 * f0;
            g.frames = a.frames + f0 * prm->H * prm->W * 3;
            g.stats = a.stats ? a.stats + f0 : nullptr;
            g.F = fc;
            g.redo = flags;
            EvArgs p = g;
            p.bin_bytes = (int)b10.bin_bytes;
            p.b10_ws = reinterpret_cast<unsigned *>(sort_base);
            p.b10_rows = b10.rows, p.b10_bands = b10.bands, p.b10_cap = b10.cap, p.b10_events = prm->max_frame_events;
            p.band_magic = (unsigned)((0x100000000ull + (unsigned)b10.rows - 1) / (unsigned)b10.rows);
            const int wgs = fc < (int)wgs_fit ? fc : (int)wgs_fit;
            hipLaunchKernelGGL(events_band10_kernel<EV>, dim3(wgs), dim3(EV_THREADS), b10_lds, hs, p);
            hipLaunchKernelGGL(events_to_frames_kernel<EV>, dim3(fc < grid ? fc : grid), dim3(EV_THREADS), lds, hs, g);
            EC_CHECK_HIP(hipGetLastError());
        }
        return EC_OK;
    }
    if (pack10_fits(prm->H, prm->W) && prm->sort_workspace && prm->sort_workspace_bytes >= 256 &&
        !events_env_off("EC_EVENTS_NO_PACK10")) {
        // whole-frame 10-bit path first, then the 32-bit kernel for the frames it flagged (none, for
        // ordinary data: its workgroups return at once).  The caller's workspace holds the flags.
        if (int rc = ec::ensure_dynamic_lds(reinterpret_cast<const void *>(events_pack10_kernel<EV>), lds))
            return rc;
        const int p10_lds = (int)pack10_bytes(prm->H, prm->W) + EV_SCRATCH_BYTES;
        const int p10_per_cu = 2 * p10_lds <= LDS_TOTAL ? 2 : 1;     // 1024 threads each: at most two fit a CU
        uint8_t *flags = static_cast<uint8_t *>(prm->sort_workspace);
        const long cap = (long)prm->sort_workspace_bytes;
        const long M2 = (long)prm->H * prm->W * 2;
        if (a.sort_ws) {            // the workspace is the flag array here, not sort slots
            a.sort_ws = nullptr;
            a.sort_cap = 0;
            grid = F;
        }
        for (long f0 = 0; f0 < F; f0 += cap) {
            const int fc = (int)(F - f0 < cap ? F - f0 : cap);
            EvArgs g = a;
            g.range = a.range + 2 * f0;
            g.frames = a.frames + f0 * prm->H * prm->W * 3;
            g.raw = a.raw ? a.raw + f0 * M2 : nullptr;
            g.kept = a.kept ? a.kept + f0 * M2 : nullptr;
            g.stats = a.stats ? a.stats + f0 : nullptr;
            g.F = fc;
            g.redo = flags;
            EvArgs p = g;
            p.bin_bytes = (int)pack10_bytes(prm->H, prm->W);
            // one persistent workgroup per CU for the 10-bit kernel (it writes every frame's flag); the 32-bit kernel
            // behind it walks the flags with one workgroup per CU as well and touches only the flagged frames
            const int p10_cus = ec::cu_count() > 0 ? ec::cu_count() : 256;
            const int wgs = fc < p10_cus ? fc : p10_cus;
            const int p10_wgs = fc < p10_cus * p10_per_cu ? fc : p10_cus * p10_per_cu;
            hipLaunchKernelGGL(events_pack10_kernel<EV>, dim3(p10_wgs), dim3(EV_THREADS), p10_lds, hs, p);
            hipLaunchKernelGGL(events_to_frames_kernel<EV>, dim3(wgs), dim3(EV_THREADS), lds, hs, g);
            EC_CHECK_HIP(hipGetLastError());
        }
        return EC_OK;
    }
    hipLaunchKernelGGL(events_to_frames_kernel<EV>, dim3(grid), dim3(EV_THREADS), lds, hs, a);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

}  // namespace

#ifdef EC_EVENTS_DIAG
extern "C" EC_API int ec_events_phase_times(unsigned long long *host, int frames)
{
    EC_REQUIRE(host && frames > 0 && frames <= 4096, "ec_events_phase_times: bad arguments");
    EC_CHECK_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ev_phase), (size_t)frames * 8 * sizeof(unsigned long long)));
    return EC_OK;
}
#endif

extern "C" EC_API size_t ec_events_sort_workspace_bytes(const ec_events_params *prm)
{
    if (!prm || prm->H <= 0 || prm->W <= 0) return 0;
    if (pack10_fits(prm->H, prm->W)) return 65536;    // per-frame redo flags of the 10-bit path
    if (prm->max_frame_events <= 0) return 0;
    {   // banded 10-bit path: the flags, then every CU's code regions
        const B10Plan b10 = b10_plan(prm->H, prm->W, prm->max_frame_events);
        const int cus_now = ec::cu_count() > 0 ? ec::cu_count() : 256;
        if (b10.ok) return (size_t)B10_FLAG_BYTES + (size_t)cus_now * b10.region_bytes;
    }
    const long row_bytes = (long)prm->W * 2 * 4;
    const long cache_bytes = ((long)prm->max_frame_events * 4 + 15) / 16 * 16;
    const long left = (long)EV_BIN_BYTES - cache_bytes;
    if (left >= row_bytes && ec::ceil_div(prm->H, (int)(left / row_bytes)) <= 8) return 0;   // LDS cache
    if (row_bytes * prm->H <= EV_BIN_BYTES) return 0;                                        // one band
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess)
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return (size_t)cus * (size_t)prm->max_frame_events * 4;
}

extern "C" EC_API int ec_events_to_frames(const float *events, const int64_t *frame_range, int F,
                                          const ec_events_params *prm, uint8_t *frames,
                                          int32_t *raw_counts, int32_t *kept_counts,
                                          ec_frame_stats *stats, ec_stream_t stream)
{
    return launch_events<float4>(events, frame_range, F, prm, frames, raw_counts, kept_counts, stats,
                                 stream);
}

extern "C" EC_API int ec_events_to_frames_packed(const uint64_t *events, const int64_t *frame_range,
                                                 int F, const ec_events_params *prm, uint8_t *frames,
                                                 int32_t *raw_counts, int32_t *kept_counts,
                                                 ec_frame_stats *stats, ec_stream_t stream)
{
    return launch_events<packed_t>(events, frame_range, F, prm, frames, raw_counts, kept_counts, stats,
                                   stream);
}

extern "C" EC_API int ec_center_events_packed(uint64_t *events, const int64_t *sample_range, int B,
                                              int H, int W, ec_stream_t stream)
{
    EC_REQUIRE(B >= 0 && H > 0 && W > 0, "ec_center_events_packed: bad arguments");
    if (B == 0) return EC_OK;
    EC_REQUIRE(events && sample_range, "ec_center_events_packed: null buffer");
    EC_REQUIRE(((uintptr_t)events & 7) == 0, "ec_center_events_packed: events must be 8-byte aligned");
    hipLaunchKernelGGL(center_packed_kernel, dim3(B), dim3(1024), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<packed_t *>(events),
                       reinterpret_cast<const long long *>(sample_range), H, W);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}

extern "C" EC_API int ec_pack_events(const float *events, int64_t n, uint64_t *packed,
                                     uint32_t *n_unrepresentable, ec_stream_t stream)
{
    EC_REQUIRE(n >= 0, "ec_pack_events: n=%lld", (long long)n);
    if (n == 0) return EC_OK;
    EC_REQUIRE(events && packed, "ec_pack_events: null buffer");
    EC_REQUIRE(((uintptr_t)events & 15) == 0 && ((uintptr_t)packed & 7) == 0,
               "ec_pack_events: events must be 16-byte and packed 8-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n_unrepresentable) EC_CHECK_HIP(hipMemsetAsync(n_unrepresentable, 0, 4, s));
    const long long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(pack_events_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                       s, reinterpret_cast<const float4 *>(events), (long long)n,
                       reinterpret_cast<packed_t *>(packed), n_unrepresentable);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}
