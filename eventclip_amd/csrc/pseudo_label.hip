// Pseudo-label selection over test-time-augmentation views (gfx950).
//
// Replaces the per-batch tensor code of gen_data.py:132-164: the classifier's aggregated
// probabilities of the V views of a sample (V = 4 with --tta: identity, h-flip, t-flip, h+t-flip,
// datasets/event2img.py:94-112; V = 1 without) are reduced to one prediction and a keep / drop
// decision:
//   * per-view argmax classes must agree when tta_consistent (:139-143),
//   * the smallest per-view top probability must exceed conf_thresh when tta_min_prob (:145-147),
//   * the view-mean distribution (:148) gives the label and its confidence (:155), which must
//     exceed conf_thresh (:156-158).
// One workgroup per sample; the V x K probabilities are read once (HBM-bound, 4 B per class-view).
// Ties resolve to the lowest class index like torch.argmax / torch.max.
#include "common.h"

namespace {

constexpr int PL_THREADS = 256;
constexpr int PL_MAX_VIEWS = 8;

struct Best {
    float v;
    int i;
};

__device__ __forceinline__ Best better(Best a, Best b)
{
    // NaN never wins; equal values keep the lower index
    if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}

__device__ Best block_best(Best b, Best *scratch)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Best t;
        t.v = __shfl_down(b.v, o, 64);
        t.i = __shfl_down(b.i, o, 64);
        b = better(b, t);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[wave] = b;
    __syncthreads();
    Best r = scratch[0];
#pragma unroll
    for (int w = 1; w < PL_THREADS / 64; w++) r = better(r, scratch[w]);
    return r;
}

__global__ __launch_bounds__(PL_THREADS) void pseudo_label_kernel(
    const float *probs, int V, int K, float conf_thresh, int consistent, int min_prob,
    float *mean_probs, int *pred, float *max_prob, unsigned char *selected)
{
    __shared__ Best scratch[PL_THREADS / 64];
    const long b = blockIdx.x;
    const float *p = probs + b * V * K;
    Best view[PL_MAX_VIEWS];
    Best mean = {-INFINITY, 0x7fffffff};
#pragma unroll
    for (int v = 0; v < PL_MAX_VIEWS; v++) view[v] = Best{-INFINITY, 0x7fffffff};
    const float inv = 1.f / (float)V;
    for (int k = threadIdx.x; k < K; k += PL_THREADS) {
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < PL_MAX_VIEWS; v++) {
            if (v < V) {
                const float x = p[(long)v * K + k];
                view[v] = better(view[v], Best{x, k});
                s += x;                                  // views summed in order, like a dim-1 reduce
            }
        }
        const float m = V == 1 ? s : s * inv;            // gen_data.py:148 (mean) / :150
        if (mean_probs) mean_probs[b * K + k] = m;
        mean = better(mean, Best{m, k});
    }
    mean = block_best(mean, scratch);
    bool ok = mean.v > conf_thresh;                      // gen_data.py:156
    float lowest = INFINITY;
    int first = -1;
#pragma unroll
    for (int v = 0; v < PL_MAX_VIEWS; v++) {
        if (v < V && V > 1 && (consistent || min_prob)) {
            const Best t = block_best(view[v], scratch);
            if (v == 0) first = t.i;
            if (consistent && t.i != first) ok = false;  // :141-143
            lowest = fminf(lowest, t.v);
        }
    }
    if (V > 1 && min_prob && !(lowest > conf_thresh)) ok = false;   // :146-147
    if (threadIdx.x == 0) {
        pred[b] = mean.i;
        max_prob[b] = mean.v;
        selected[b] = ok ? 1 : 0;
    }
}

}  // namespace

extern "C" EC_API int ec_pseudo_label(const float *probs, int B, int V, int K, float conf_thresh,
                                      int tta_consistent, int tta_min_prob, float *mean_probs,
                                      int32_t *pred, float *max_prob, uint8_t *selected,
                                      ec_stream_t stream)
{
    EC_REQUIRE(B >= 0 && K > 0, "ec_pseudo_label: bad shape B=%d K=%d", B, K);
    EC_REQUIRE(V >= 1 && V <= PL_MAX_VIEWS, "ec_pseudo_label: V=%d views (1..%d)", V, PL_MAX_VIEWS);
    if (B == 0) return EC_OK;
    EC_REQUIRE(probs && pred && max_prob && selected, "ec_pseudo_label: null buffer");
    hipLaunchKernelGGL(pseudo_label_kernel, dim3(B), dim3(PL_THREADS), 0,
                       static_cast<hipStream_t>(stream), probs, V, K, conf_thresh, tta_consistent,
                       tta_min_prob, mean_probs, pred, max_prob, selected);
    EC_CHECK_HIP(hipGetLastError());
    return EC_OK;
}
