// Event-space training augmentation on the device (gfx950).
//
// Replaces NCaltech101._augment_events (datasets/caltech.py:153-163) = the three functions of
// datasets/utils.py applied per training sample, in this order:
//   random_time_flip_events   (:26-35, N-ImageNet only: reverse the order, t -> t[0] - t, p -> -p)
//   random_shift_events       (:4-15: x += dx, y += dy, then DROP every event that left the sensor)
//   random_flip_events_along_x (:18-23: x -> W - 1 - x)
// The random draws stay on the host (same numpy calls in the same order, eventclip_amd/augment.py);
// this kernel applies them.  Dropping events changes the event count and with it every later chunk
// boundary of split_event_count, so the survivors are compacted in order: one workgroup per sample walks
// the (possibly reversed) stream in coalesced tiles of 1024 events, a ballot + the waves' counts place every
// survivor.  HBM-bound: 16 B in + <= 16 B out per event.
#include "common.h"

namespace {

constexpr int AUG_THREADS = 1024;

__global__ __launch_bounds__(AUG_THREADS) void augment_events_kernel(const float4 *events, const long long *range,
                                                                     const int *params, int H, int W, float4 *out,
                                                                     long long *counts)
{
    __shared__ int wave_cnt[2][AUG_THREADS / 64];
    const int b = blockIdx.x;
    const long long e0 = range[2 * b], n = range[2 * b + 1] - e0;
    const float4 *ev = events + e0;
    float4 *dst = out + e0;
    const int dx = params[4 * b], dy = params[4 * b + 1], flip_x = params[4 * b + 2], flip_t = params[4 * b + 3];
    const float t_last = n > 0 ? ev[n - 1].z : 0.f;        // events[0, 2] of the reversed stream (utils.py:32)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // The stream is walked in tiles of 1024 consecutive events, one per thread (coalesced, also when the time flip
    // reverses it); survivors keep their order: a ballot gives the position inside the wave, the waves' counts go
    // through LDS (double-buffered: one barrier per tile), the running total carries over to the next tile.
    long long done = 0;
    int buf = 0;
    for (long long t0 = 0; t0 < n; t0 += AUG_THREADS, buf ^= 1) {
        const long long i = t0 + threadIdx.x;            // index in the stream AFTER the optional time flip
        bool ok = false;
        float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n) {
            e = ev[flip_t ? n - 1 - i : i];
            if (flip_t) e.z = t_last - e.z, e.w = -e.w;
            e.x += (float)dx, e.y += (float)dy;                                        // utils.py:8-9
            ok = e.x >= 0.f && e.x < (float)W && e.y >= 0.f && e.y < (float)H;         // :11-12
            if (flip_x) e.x = (float)(W - 1) - e.x;                                    // :22
        }
        const unsigned long long mask = __ballot(ok);
        const int before = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[buf][wave] = __popcll(mask);
        __syncthreads();
        int base = 0, total = 0;
#pragma unroll
        for (int w = 0; w < AUG_THREADS / 64; w++) {
            const int c = wave_cnt[buf][w];
            base += w < wave ? c : 0;
            total += c;
        }
        if (ok) dst[done + base + before] = e;
        done += total;
    }
    if (threadIdx.x == 0) counts[b] = done;
}

}  // namespace

extern "C" EC_API int ec_augment_events(const float *events, const int64_t *sample_range, int B,
                                        const int32_t *params, int H, int W, float *events_out,
                                        int64_t *counts_out, ec_stream_t stream)
{
    EC_REQUIRE(B >= 0 && H > 0 && W > 0, "ec_augment_events: bad arguments");
    if (B == 0) return EC_OK;
    EC_REQUIRE(events && sample_range && params && events_out && counts_out, "ec_augment_events: null buffer");
    EC_REQUIRE(events != events_out, "ec_augment_events: in-place operation is not supported");
    EC_REQUIRE((((uintptr_t)events | (uintptr_t)events_out) & 15) == 0, "ec_augment_events: 16-byte alignment");
    hipLaunchKernelGGL(augment_events_kernel, dim3(B), dim3(AUG_THREADS), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4 *>(events), reinterpret_cast<const long long *>(sample_range),
                       params, H, W, reinterpret_cast<float4 *>(events_out),
                       reinterpret_cast<long long *>(counts_out));
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}
