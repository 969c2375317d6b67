/*
 * eventclip_hip.h -- C ABI of libeventclip_hip.so (MI355X / gfx950 only).
 *
 * The reference (Wuziyi616/EventCLIP) is pure Python with no FFI: its seam is
 * duck-typed Python.  This header is the native boundary a
 * maintainer binds instead; INTEGRATION.md shows the ctypes stubs.  Every entry
 * point takes plain device pointers and sizes plus the HIP stream to enqueue
 * on; nothing here allocates, synchronises or touches torch.  All functions
 * return 0 on success and a negative EC_ERR_* code otherwise (message via
 * ec_last_error()).
 *
 * Each block cites the reference interface it replaces.
 */
#ifndef EVENTCLIP_HIP_H
#define EVENTCLIP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EC_API __attribute__((visibility("default")))

typedef void *ec_stream_t; /* hipStream_t */

enum {
    EC_OK = 0,
    EC_ERR_INVALID = -1,   /* bad argument (the reference would assert / raise) */
    EC_ERR_HIP = -2,       /* a HIP runtime call failed */
    EC_ERR_UNSUPPORTED = -3,
    EC_ERR_WORKSPACE = -4, /* workspace too small */
};

/* 16-bit storage type of activations / GEMM operands */
enum { EC_F16 = 0, EC_BF16 = 1 };

EC_API const char *ec_last_error(void);
EC_API int ec_version(void);
/* number of CUs / name of the current device, for reports */
EC_API int ec_device_info(int *cu_count, char *name, int name_len);

/* ------------------------------------------------------------------------
 * events -> histogram frames.
 * Replaces datasets/vis.py: events2frames (:75-117) = parse_events (:44-52) +
 * per-chunk make_event_histogram (:6-41).  The chunk bounds
 * (split_event_count, :55-72) are computed by the caller from event counts
 * alone and handed in as [begin, end) pairs, so overlapping chunks (:67-69)
 * need no copy.
 * ------------------------------------------------------------------------ */
typedef struct {
    uint64_t sum;        /* sum of counts over both channels (events with p != 0 inside the sensor) */
    uint64_t sumsq;      /* sum of squared counts */
    uint32_t nnz;        /* bins with count > 0 */
    uint32_t max_kept;   /* max count after hot-pixel removal (vis.py:27) */
    uint32_t dropped;    /* events outside the sensor (the reference raises, vis.py:11) */
    uint32_t ambiguous;  /* bins whose count is within 1e-9 relative of the threshold: the only
                            place where the integer-exact threshold can differ from numpy's */
    double thr;          /* thresh * std + mean (vis.py:24); NaN if the population is empty */
} ec_frame_stats;

typedef struct {
    int H, W;                 /* sensor resolution (shape=, vis.py:79) */
    double thresh;            /* hot-pixel threshold in std units, <= 0 disables (vis.py:17) */
    int count_non_zero;       /* statistics over non-zero bins only (vis.py:18-20) */
    int background_mask;      /* alpha-blend onto white (vis.py:34-37) */
    uint8_t red[3], blue[3];  /* colour of the positive / negative channel (vis.py:95-104) */
} ec_events_params;

/*
 * events:      float32 [n_events_total, 4] rows (x, y, t, p), device.
 * frame_range: int64 [F, 2] (begin, end) event-row indices per frame, device.
 * frames:      uint8 [F, H, W, 3], device (out).
 * raw_counts / kept_counts: optional int32 [F, H, W, 2] (out): counts before /
 *              after hot-pixel removal (debug outputs for the bit-exact tests).
 * stats:       optional ec_frame_stats [F] (out).
 */
EC_API int ec_events_to_frames(const float *events, const int64_t *frame_range, int F,
                               const ec_events_params *prm, uint8_t *frames, int32_t *raw_counts,
                               int32_t *kept_counts, ec_frame_stats *stats, ec_stream_t stream);

/* ------------------------------------------------------------------------
 * 16-bit MFMA GEMM with fused epilogue: C[M,N] = epi(A[M,K] . W[N,K]^T + bias).
 * The building block behind every nn.Linear / in_proj / out_proj / conv1 /
 * projection of the CLIP towers the reference calls through
 * clip_model.encode_image / encode_text (models/clip_cls.py:84,101; module
 * structure from un-vendored openai/CLIP clip/model.py).  Exposed for unit
 * tests and micro-benchmarks; ec_vit_encode / ec_text_encode drive it.
 * ------------------------------------------------------------------------ */
enum {
    EC_EPI_STORE16 = 0, /* C16 = acc + bias */
    EC_EPI_GELU16 = 1,  /* C16 = QuickGELU(acc + bias), x * sigmoid(1.702 x) */
    EC_EPI_RESID32 = 2, /* C32 += acc + bias   (fp32 residual stream, in place) */
    EC_EPI_STORE32 = 3, /* C32 = acc + bias */
};

typedef struct {
    int M, N, K;       /* K % 64 == 0, N % 16 == 0 */
    int dtype;         /* EC_F16 / EC_BF16: A, W and 16-bit outputs */
    int epilogue;      /* EC_EPI_* */
    int variant;       /* 0 = default tiling; others select tilings for A/B runs */
    const void *A;     /* [M, K] 16-bit, row stride lda elements (0 = K) */
    long lda;
    const void *W;     /* [N, K] 16-bit, dense (nn.Linear weight layout) */
    const float *bias; /* [N] fp32 or NULL */
    void *C;           /* [M, N] 16-bit or fp32 by epilogue, row stride ldc (0 = N) */
    long ldc;
} ec_gemm_args;

EC_API int ec_gemm(const ec_gemm_args *args, ec_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EVENTCLIP_HIP_H */
