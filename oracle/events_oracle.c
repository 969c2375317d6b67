/*
 * oracle/events_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference's events -> histogram-frame path
 * (/root/reference/datasets/vis.py).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this; the product path
 * (eventclip_amd/) never does.
 *
 * Pinned: tests/golden/events_*.npz were produced by importing the reference's
 * own vis.py in the build container (tools/make_golden_events.py, numpy 2.2.6,
 * so the float stage runs in float64, see make_event_histogram below) and this
 * file reproduces every one of them bit for bit.
 *
 * Every function cites the reference lines it follows.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EXPORT __attribute__((visibility("default")))

/* numpy's float64 add.reduce over one contiguous run: pairwise summation with
 * 8 interleaved accumulators per <=128-element block (numpy
 * _core/src/umath/loops_utils.h.src, DOUBLE_pairwise_sum).  ndarray.std()
 * (vis.py:19,22) sums the squared deviations through this, so the oracle has
 * to as well to reproduce the hot-pixel threshold to the last bit. */
static double np_pairwise_sum(const double *a, long n)
{
    if (n < 8) {
        double res = 0.;
        for (long i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        long i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
    }
}

/* ndarray.sum() drives that inner loop through a buffered iterator: the run is
 * cut into getbufsize() = 8192-element pieces whose pairwise sums are added to
 * the accumulator in order (checked against numpy 2.2.6 in the build
 * container for n up to 614400: tools/make_golden_events.py asserts it). */
static double np_sum_f64(const double *a, long n)
{
    double acc = 0.;
    for (long i = 0; i < n; i += 8192)
        acc += np_pairwise_sum(a + i, (n - i < 8192) ? (n - i) : 8192);
    return acc;
}

EXPORT double ec_oracle_np_sum(const double *a, long n) { return np_sum_f64(a, n); }

/* ndarray.mean()/.std() of an int64 population (numpy _core/_methods.py _mean,
 * _var): mean = float64(sum)/count (integer sum is exact); var = pairwise_sum(
 * (x-mean)^2)/count; std = sqrt(var).  Empty population -> NaN, as numpy. */
static void np_mean_std(const int64_t *v, long n, double *mean, double *std, double *scratch)
{
    if (n == 0) { *mean = NAN; *std = NAN; return; }
    int64_t s = 0;
    for (long i = 0; i < n; i++) s += v[i];
    double m = (double)s / (double)n;
    for (long i = 0; i < n; i++) { double d = (double)v[i] - m; scratch[i] = d * d; }
    double ss = np_sum_f64(scratch, n);
    *mean = m;
    *std = sqrt(ss / (double)n);
}

/*
 * make_event_histogram, vis.py:6-41.
 *   x,y,p: int32 event columns of ONE chunk (already truncated, vis.py:50).
 *   red/blue: uint8[3] colour of the positive / negative channel (vis.py:95-104).
 *   raw (optional): int64[H*W*2] counts before hot-pixel removal (vis.py:10-14)
 *   kept (optional): int64[H*W*2] counts after hot-pixel removal (vis.py:17-24)
 *   img: uint8[H*W*3].
 * Returns 0, or -1 if an event lies outside the sensor (the reference's
 * bincount/reshape would raise there, vis.py:11).
 *
 * Float stage: under numpy >= 2 `hist.astype(float32) / hist.max()` promotes to
 * float64 (NEP 50; the reference's pinned numpy 1.25 kept float32), the
 * `hist @ cmap` matmul goes to OpenBLAS dgemm whose K=2 inner product is
 * fma(b, c1, a*c0), and every later step is a separate float64 ufunc
 * (one rounding each).  np.round is round-half-to-even.  All-zero frames give
 * 0/0 = NaN, which astype(uint8) turns into 0 on x86.
 */
/* Float stage of vis.py:27-39: 0 = float64 (numpy >= 2, NEP 50: this image, and what the committed
 * fixtures were generated under), 1 = float32 (the reference's pinned numpy 1.25.2,
 * environment.yml:49: `float32 array / np.int64 scalar` stays float32 under value-based casting, the
 * matmul is sgemm, and every later ufunc runs in float32 because python floats do not upcast). */
static int g_float32_stage = 0;
EXPORT void ec_oracle_set_float32_stage(int on) { g_float32_stage = on; }

EXPORT int ec_oracle_event_histogram(const int32_t *x, const int32_t *y, const int32_t *p, long n,
                                     int H, int W, double thresh, int count_non_zero,
                                     int background_mask, const uint8_t *red, const uint8_t *blue,
                                     int64_t *raw, int64_t *kept, uint8_t *img)
{
    const long M = (long)H * W * 2;
    int64_t *hist = (int64_t *)calloc((size_t)M, sizeof(int64_t));
    double *scratch = (double *)malloc((size_t)M * sizeof(double));
    int64_t *sel = (int64_t *)malloc((size_t)M * sizeof(int64_t));
    int rc = 0;

    /* vis.py:10-14: two bincounts over x + y*W, stacked on the last axis */
    for (long i = 0; i < n; i++) {
        if (p[i] == 0) continue;
        if (x[i] < 0 || x[i] >= W || y[i] < 0 || y[i] >= H) { rc = -1; goto done; }
        hist[((long)y[i] * W + x[i]) * 2 + (p[i] > 0 ? 0 : 1)] += 1;
    }
    if (raw) memcpy(raw, hist, (size_t)M * sizeof(int64_t));

    /* vis.py:17-24: hot-pixel removal */
    if (thresh > 0) {
        double mean, std;
        if (count_non_zero) {
            long k = 0;
            for (long i = 0; i < M; i++) if (hist[i] > 0) sel[k++] = hist[i];
            np_mean_std(sel, k, &mean, &std, scratch);
        } else {
            np_mean_std(hist, M, &mean, &std, scratch);
        }
        double thr = thresh * std + mean;
        for (long i = 0; i < M; i++) if ((double)hist[i] > thr) hist[i] = 0;
    }
    if (kept) memcpy(kept, hist, (size_t)M * sizeof(int64_t));

    /* vis.py:27: normalise by the max of what is left */
    int64_t mx = 0;
    for (long i = 0; i < M; i++) if (hist[i] > mx) mx = hist[i];
    const double dmx = (double)mx;

    if (g_float32_stage) {
        const float fmx = (float)mx;                 /* the int64 scalar is cast to the array's float32 */
        for (long q = 0; q < (long)H * W; q++) {
            float a = (float)hist[2 * q] / fmx;
            float b = (float)hist[2 * q + 1] / fmx;
            float w = 0.f;
            if (background_mask) {
                w = a + b;
                if (w < 0.f) w = 0.f;
                if (w > 1.f) w = 1.f;
            }
            for (int c = 0; c < 3; c++) {
                float v = fmaf(b, (float)blue[c], a * (float)red[c]);   /* sgemm, K = 2 */
                if (background_mask) {
                    float t1 = v * w;
                    float t2 = 255.f * (1.f - w);
                    v = t1 + t2;
                }
                float r = nearbyintf(v);
                img[3 * q + c] = isnan(r) ? 0 : (uint8_t)r;
            }
        }
        goto done;
    }
    for (long q = 0; q < (long)H * W; q++) {
        double a = (double)(float)hist[2 * q] / dmx;     /* astype(float32) is exact below 2^24 */
        double b = (double)(float)hist[2 * q + 1] / dmx;
        double w = 0.;
        if (background_mask) {
            /* vis.py:35: clip(hist.sum(-1), 0, 1); np.clip = minimum(maximum(x, 0), 1), NaN-propagating */
            w = a + b;
            if (w < 0.) w = 0.;
            if (w > 1.) w = 1.;
        }
        for (int c = 0; c < 3; c++) {
            /* vis.py:30-31 */
            double v = fma(b, (double)(float)blue[c], a * (double)(float)red[c]);
            if (background_mask) {
                /* vis.py:36-37 */
                double t1 = v * w;
                double t2 = 255. * (1. - w);
                v = t1 + t2;
            }
            /* vis.py:39 */
            double r = nearbyint(v);
            img[3 * q + c] = isnan(r) ? 0 : (uint8_t)r;
        }
    }
done:
    free(hist); free(scratch); free(sel);
    return rc;
}

/* split_event_count, vis.py:55-72 (the t0/t1 it also returns are unused by
 * events2frames).  Writes chunk bounds into idx0/idx1 (capacity cap) and
 * returns the number of chunks. */
EXPORT long ec_oracle_split_event_count(long tot_cnt, long N, long *idx0, long *idx1, long cap)
{
    long f = 0;
    if (tot_cnt < N) {                       /* vis.py:60-61 */
        if (cap > 0) { idx0[0] = 0; idx1[0] = tot_cnt; }
        return 1;
    }
    /* vis.py:64-65: idx = arange(0, tot, N); chunks are consecutive pairs, so a
     * start s is kept iff its successor s + N is still in the arange */
    long last = ((tot_cnt - 1) / N) * N;     /* idx[-1] */
    for (long s = 0; s + N < tot_cnt; s += N) {
        if (f < cap) { idx0[f] = s; idx1[f] = s + N; }
        f++;
    }
    if ((double)(tot_cnt - last) > (double)N * 0.5) {   /* vis.py:67-69 */
        if (f < cap) { idx0[f] = tot_cnt - N; idx1[f] = tot_cnt; }
        f++;
    }
    return f;
}

/*
 * events2frames, vis.py:75-117, for float32 [n_ev,4] (x,y,t,p) input as the
 * dataset readers hand it over (caltech.py:151).  parse_events (vis.py:44-52)
 * truncates x,y,p to int32; t only feeds the unused t0/t1.
 *   frames: uint8[F*H*W*3]; raw/kept optional int64[F*H*W*2].
 * Returns F (number of frames), or -1 on an out-of-sensor event, or -2 if
 * max_frames is too small.
 */
EXPORT long ec_oracle_events2frames(const float *ev, long n_ev, long N, int H, int W, double thresh,
                                    int count_non_zero, int background_mask, const uint8_t *red,
                                    const uint8_t *blue, long max_frames, uint8_t *frames,
                                    int64_t *raw, int64_t *kept)
{
    long *idx0 = (long *)malloc(sizeof(long) * (size_t)(max_frames + 2));
    long *idx1 = (long *)malloc(sizeof(long) * (size_t)(max_frames + 2));
    long F = ec_oracle_split_event_count(n_ev, N, idx0, idx1, max_frames);
    long rc = F;
    if (F > max_frames) { rc = -2; goto out; }
    int32_t *x = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_ev + 1));
    int32_t *y = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_ev + 1));
    int32_t *p = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_ev + 1));
    for (long i = 0; i < n_ev; i++) {        /* vis.py:50 */
        x[i] = (int32_t)ev[4 * i + 0];
        y[i] = (int32_t)ev[4 * i + 1];
        p[i] = (int32_t)ev[4 * i + 3];
    }
    const long M = (long)H * W;
    for (long f = 0; f < F; f++) {           /* vis.py:106-115 */
        long i0 = idx0[f], cnt = idx1[f] - idx0[f];
        int r = ec_oracle_event_histogram(x + i0, y + i0, p + i0, cnt, H, W, thresh, count_non_zero,
                                          background_mask, red, blue,
                                          raw ? raw + f * M * 2 : 0, kept ? kept + f * M * 2 : 0,
                                          frames + f * M * 3);
        if (r != 0) { rc = -1; break; }
    }
    free(x); free(y); free(p);
out:
    free(idx0); free(idx1);
    return rc;
}
