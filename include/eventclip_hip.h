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

/* 16-bit storage type of activations / GEMM operands.
 * EC_F16 is the default and the type the accuracy contract is stated for: it is the reference's own GPU dtype
 * (clip.load(arch, 'cuda'), /root/reference/test.py:25-26), and on it the towers' logits stay within 1e-3 of the fp32
 * oracle relative to max |logit| on weights at OpenAI's init scales, and closer to fp32 than the reference's fp16
 * arithmetic on input-dependent weights (tests/test_configs_gpu.py).
 * EC_BF16 (the same MFMA rate, 8 fewer mantissa bits) is OUTSIDE that contract: 2.8e-3 .. 4.6e-3 on the same
 * weights; every kernel supports it and is tested at bf16's own tolerances (1e-2). */
enum { EC_F16 = 0, EC_BF16 = 1 };

EC_API const char *ec_last_error(void);

/* ABI version of THIS header; ec_version() returns the one the loaded library was built from.
 *   100  rounds 1 - 5 (the number was never bumped; ec_classify changed its argument list twice and ec_gemm_args /
 *        ec_vit_weights grew in that time -- a caller built against any of those headers must be rebuilt)
 *   600  round 6: ec_classify -> ec_classify_prep_text + ec_classify_v2; ec_abi_check
 * Structs are passed by pointer and only ever grow AT THE END; a library reads every field of ITS OWN struct
 * definition, so a caller built against an older (shorter) struct would have the tail read from past its object.
 * ec_abi_check(EC_ABI_VERSION, sizeof ...) -- EC_ABI_CHECK() below -- compares the caller's header version and struct
 * sizes with the library's and returns EC_ERR_INVALID (message in ec_last_error()) on any difference: call it once
 * after loading the library, before any other entry point.  (eventclip_amd/_lib.py does, with its ctypes mirrors.) */
#define EC_ABI_VERSION 600
EC_API int ec_version(void);
EC_API int ec_abi_check(int header_version, size_t gemm_args_bytes, size_t block_weights_bytes, size_t vit_weights_bytes,
                        size_t text_weights_bytes, size_t events_params_bytes, size_t adapter_weights_bytes);
#define EC_ABI_CHECK()                                                                                              \
    ec_abi_check(EC_ABI_VERSION, sizeof(ec_gemm_args), sizeof(ec_block_weights), sizeof(ec_vit_weights),            \
                 sizeof(ec_text_weights), sizeof(ec_events_params), sizeof(ec_adapter_weights))
/* number of CUs / name of the current device, for reports */
EC_API int ec_device_info(int *cu_count, char *name, int name_len);

/* Launch profiler: between begin and end every kernel launch this library makes is
 * bracketed by hipEvents on its own stream; end() waits for them and returns one
 * entry per kernel symbol (launch count, summed duration, algorithmic flops / bytes
 * as the call sites state them).  bench.py's `roofline` comes from here. */
typedef struct {
    char name[64];
    long launches;
    double total_ms;
    double flops; /* algorithmic FLOPs (2*M*N*K per GEMM launch, ...) */
    double bytes; /* algorithmic HBM bytes where the call site states them, else 0 */
} ec_profile_entry;
EC_API int ec_profile_begin(void);
EC_API int ec_profile_end(ec_profile_entry *out, int cap, int *n_out);

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
    int max_frame_events;     /* upper bound of events per frame (N of split_event_count), 0 = unknown;
                                 lets the kernel keep a frame's events in LDS and read HBM once */
    int flip_x;               /* test-time augmentation: x -> W - 1 - x (datasets/utils.py:18-23) */
    int negate_p;             /* test-time augmentation: p -> -p; together with frame ranges taken on
                                 the reversed event order this is the time flip of utils.py:26-35 */
    void *sort_workspace;     /* optional device scratch of ec_events_sort_workspace_bytes().  Sensors whose
                                 histogram fits a CU as 10-bit counts (up to ~59 000 pixels: N-Caltech, N-Cars):
                                 64 KiB of per-frame flags for the whole-frame kernel.  Larger sensors
                                 (N-ImageNet) with max_frame_events set: the flags plus, per CU, a region of
                                 band-local bin codes per (row band, wave) sized for the worst case -- every
                                 event in one band -- so that the one scan that reads the events from HBM also
                                 routes them to their bands (480 x 640, 70 000 events: 440 MB for 256 CUs).
                                 Without it (or with raw / kept counts requested) the 32-bit band kernel of
                                 rounds 1-2 runs: same results, more passes */
    size_t sort_workspace_bytes;
    int float32_stage;        /* 0: the float stage of vis.py:27-39 in float64, as numpy >= 2 runs the
                                 reference (NEP 50); != 0: in float32, as the reference's pinned numpy
                                 1.25.2 ran it (environment.yml:49).  The two differ by 1 LSB at exact
                                 .5 ties only (e.g. 127 * 3/6); counts are identical. */
    int64_t total_events;     /* sum of (end - begin) over frame_range if the caller knows it, else 0:
                                 only states the launch's algorithmic bytes to ec_profile_end */
} ec_events_params;

/* bytes of sort_workspace that pay off for this geometry / max_frame_events on the current device (0: not needed) */
EC_API size_t ec_events_sort_workspace_bytes(const ec_events_params *prm);

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

/* center_events (datasets/utils.py:38-57, applied to every N-Caltech / N-ImageNet sample at
 * caltech.py:176), in place: t -= min t, x -= ((x_max + x_min + 1) - W) // 2, y likewise.
 * sample_range: int64 [B, 2] (begin, end) event rows per sample. */
EC_API int ec_center_events(float *events, const int64_t *sample_range, int B, int H, int W,
                            ec_stream_t stream);

/* Training-time event augmentation (NCaltech101._augment_events, datasets/caltech.py:153-163 =
 * datasets/utils.py random_time_flip_events, random_shift_events, random_flip_events_along_x in that
 * order) applied with host-drawn parameters: params int32 [B, 4] = (x_shift, y_shift, flip_x, flip_t)
 * per sample.  Events shifted off the sensor are dropped (utils.py:11-13) and the survivors compacted
 * in order into events_out at the sample's original offset; counts_out int64 [B] receives how many. */
EC_API int ec_augment_events(const float *events, const int64_t *sample_range, int B,
                             const int32_t *params, int H, int W, float *events_out,
                             int64_t *counts_out, ec_stream_t stream);

/* ---- packed event ingest ------------------------------------------
 * The reference keeps events as float32 [n, 4] (x, y, t, p) (datasets/caltech.py:149-151; N-ImageNet's
 * structured x, y, t [us], p [0/1] is converted to that at datasets/imagenet.py:8-27) and only ever
 * reads int(x), int(y), sign(p) from them (datasets/vis.py:44-52).  The packed form carries exactly
 * that in 8 bytes, so the upload and the binning kernel's HBM reads are halved:
 *   bits  0..15  x   (parse_events' truncated integer)
 *   bits 16..31  y
 *   bits 32..33  polarity code: 0 = p == 0 (binned nowhere, vis.py:10,12), 1 = p > 0, 2 = p < 0
 *   bits 34..63  t in microseconds (30 bits)
 * Events with non-integral coordinates cannot be packed (the float path flips x before truncating,
 * utils.py:22). */
#define EC_PACKED_X(e) ((uint32_t)((e) & 0xffffu))
#define EC_PACKED_Y(e) ((uint32_t)(((e) >> 16) & 0xffffu))
#define EC_PACKED_P(e) ((uint32_t)(((e) >> 32) & 3u))
#define EC_PACKED_T(e) ((uint32_t)((e) >> 34))

/* float32 [n, 4] -> packed [n] on the device.  n_unrepresentable (optional, device uint32): number
 * of events whose coordinates are not integers in [0, 65535]; those get polarity code 0. */
EC_API int ec_pack_events(const float *events, int64_t n, uint64_t *packed,
                          uint32_t *n_unrepresentable, ec_stream_t stream);

/* ec_events_to_frames on packed events; every other argument as above. */
EC_API int ec_events_to_frames_packed(const uint64_t *events, const int64_t *frame_range, int F,
                                      const ec_events_params *prm, uint8_t *frames,
                                      int32_t *raw_counts, int32_t *kept_counts,
                                      ec_frame_stats *stats, ec_stream_t stream);

/* ec_center_events on packed events (integer form of utils.py:53-54; t relative to the sample's
 * first event).  A coordinate shifted below 0 wraps above 32767 and is dropped by the binning. */
EC_API int ec_center_events_packed(uint64_t *events, const int64_t *sample_range, int B, int H, int W,
                                   ec_stream_t stream);

/* ------------------------------------------------------------------------
 * CLIP image preprocess: uint8 frames -> model input.
 * Replaces `self.transforms(img)` per frame (datasets/event2img.py:119-122,
 * transforms = clip.load's preprocess, test.py:26-29): torchvision
 * Resize(n_px, BICUBIC) + CenterCrop(n_px) + ToTensor + Normalize over PIL.
 * The plan holds Pillow's fixed-point bicubic coefficients for one geometry.
 * ------------------------------------------------------------------------ */
enum {
    EC_PRE_CHW_F32 = 0,   /* float32 [F, 3, n_px, n_px]: the reference's tensor */
    EC_PRE_PATCHES16 = 1, /* 16-bit [F, G, kpad] im2col rows for ec_vit_encode: every value v of the
                           * (c, i, j)-ordered patch as hi = round16(v) and lo = round16(v - hi),
                           * row = [hi (3 p^2) | lo (3 p^2) | 0 ...], kpad >= 6 p^2 */
    EC_PRE_HWC_U8 = 2,    /* uint8 [F, n_px, n_px, 3]: resized + cropped, before ToTensor */
};

EC_API size_t ec_preprocess_plan_bytes(int in_h, int in_w, int n_px);
/* fills a HOST buffer; the caller uploads a copy and passes both to ec_preprocess */
EC_API int ec_preprocess_plan(int in_h, int in_w, int n_px, void *host_plan, size_t cap);
EC_API int ec_preprocess(const uint8_t *frames, int F, const void *plan_host, const void *plan_dev,
                         void *out, int mode, int patch, int kpad, int dtype, ec_stream_t stream);
/* float32 [N, 3, n_px, n_px] (what the reference feeds encode_image) -> 16-bit
 * im2col rows [N, G, kpad] in the EC_PRE_PATCHES16 layout ([hi | lo | 0], (c, i, j) order). */
EC_API int ec_patchify(const float *img, int n_img, int n_px, int patch, int kpad, void *out16,
                       int dtype, ec_stream_t stream);

/* ------------------------------------------------------------------------
 * RandAugment on uint8 frames (training only).  Replaces `self.augmentation(imgs)`
 * (datasets/event2img.py:120-121) = datasets/augment.py RandAugment.forward
 * (:159-193): the SAME op list for every view of a sample, each op one of the 14
 * of _apply_op (:10-87), which the reference runs on PIL images through
 * torchvision's functional_pil.  Bit-exact with Pillow (csrc/randaugment.hip).
 * The caller samples the ops (augment.py:142-157, host RNG) and prepares one
 * descriptor per (frame, step):
 *   AFFINE       m[6] = the output->input coefficients Image.transform(AFFINE)
 *                receives (ShearX/Y, TranslateX/Y, Rotate; BICUBIC, fill colour outside)
 *   ROT180/90/270  PIL's rotate fast paths (90 / 270 only on square frames)
 *   BRIGHTNESS / COLOR / CONTRAST / SHARPNESS   alpha = 1 + magnitude (ImageEnhance factor)
 *   POSTERIZE    param = bits kept;  SOLARIZE  param = threshold
 *   AUTOCONTRAST / EQUALIZE / IDENTITY          no parameter
 * ------------------------------------------------------------------------ */
enum {
    EC_AUG_IDENTITY = 0, EC_AUG_AFFINE = 1, EC_AUG_ROT180 = 2, EC_AUG_ROT90 = 3, EC_AUG_ROT270 = 4,
    EC_AUG_BRIGHTNESS = 5, EC_AUG_COLOR = 6, EC_AUG_CONTRAST = 7, EC_AUG_SHARPNESS = 8,
    EC_AUG_POSTERIZE = 9, EC_AUG_SOLARIZE = 10, EC_AUG_AUTOCONTRAST = 11, EC_AUG_EQUALIZE = 12,
};
typedef struct {
    int kind;      /* EC_AUG_* */
    float alpha;   /* ImageEnhance factor (C float, as Image.blend takes it) */
    double param;  /* posterize bits / solarize threshold */
    double m[6];   /* affine coefficients */
} ec_aug_op;

EC_API size_t ec_randaugment_workspace_bytes(int F, int H, int W, int num_ops);
/* frames_in / frames_out: uint8 [F, H, W, 3] (distinct buffers); ops: DEVICE array [F, num_ops]
 * (frames of one sample carry the same descriptors); fill: the RandAugment fill colour, host. */
EC_API int ec_randaugment(const uint8_t *frames_in, uint8_t *frames_out, int F, int H, int W,
                          const ec_aug_op *ops, int num_ops, const uint8_t fill[3], void *workspace,
                          size_t workspace_bytes, ec_stream_t stream);

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
    /* training (ec_vit_train_*; default variant only): */
    EC_EPI_GELU16_SAVE = 4, /* C16 = QuickGELU(acc + bias) and aux16 = acc + bias (kept for the backward pass) */
    EC_EPI_GELU_BWD16 = 5,  /* C16 = (acc + bias) * QuickGELU'(aux16): the gradient through the activation */
    /* LayerNorm folded into the GEMMs around it (the image tower's blocks; default variant only):
     * the residual stream is kept as two 16-bit planes x = hi + lo -- hi in `dtype`, lo in fp16, the same 4 bytes per
     * element as fp32 and ~2^-22 relative -- and hi, the RAW row, is the A operand of the GEMM that follows.  That
     * GEMM runs on the gamma-scaled weight W' = W diag(gamma) and finishes LayerNorm in its epilogue:
     *   LN(x) W^T + b = rstd (x W'^T) - rstd mean colsum(W') + (b + W beta),
     * with (rstd, -rstd mean) per row from ec_row_stats and colsum / the folded bias per column from the packer. */
    EC_EPI_RESID_HL = 6,    /* C = hi plane, aux = lo plane (fp16), both [M, N] at stride ldc: (hi, lo) <- split(hi + lo + acc + bias) */
    EC_EPI_STORE16_LN = 7,  /* C16 = row_stats[m][0] * acc + row_stats[m][1] * col_sums[n] + bias[n] */
    EC_EPI_GELU16_LN = 8,   /* C16 = QuickGELU(the same) */
};

typedef struct {
    int M, N, K;       /* K % 64 == 0, N % 16 == 0 */
    int dtype;         /* EC_F16 / EC_BF16: A, W and 16-bit outputs */
    int epilogue;      /* EC_EPI_* */
    int variant;       /* 0 = the kernel; anything else is EC_ERR_INVALID.  (The tilings it grew out of and the
                          stamp / timeline variants exist in the -DEC_GEMM_DIAG build for tools/ only.) */
    const void *A;     /* [M, K] 16-bit, row stride lda elements (0 = K).  Row strides (lda, ldw, ldc) are multiples of
                          8 elements below 2^21: a tile's rows are addressed with 32-bit byte offsets from the tile's
                          origin (buffer descriptors), EC_ERR_INVALID otherwise */
    long lda;
    const void *W;     /* [N, K] 16-bit, dense (nn.Linear weight layout) */
    const float *bias; /* [N] fp32 or NULL */
    void *C;           /* [M, N] 16-bit or fp32 by epilogue, row stride ldc (0 = N) */
    long ldc;
    void *diag;        /* NULL.  (Only a -DEC_GEMM_DIAG build of the library reads it: device buffer
                          for the s_memtime records of its stamp / timeline variants.) */
    /* The rest serves the training path (variant 0 only); all zero = the plain GEMM above. */
    long ldw;          /* row stride of W in elements (0 = K) */
    const float *resid;/* EC_EPI_RESID32: C = resid + acc + bias with resid [M, N] at stride ldc (NULL = C, in place) */
    void *aux;         /* 16-bit [M, N] at stride ldc: second output of GELU16_SAVE / input of GELU_BWD16 */
    int splits;        /* > 1: that many independent products over consecutive K-column ranges of A and W
                          (batch s reads columns s*K .. (s+1)*K - 1 and writes C + s * split_stride elements):
                          the partial sums of a weight gradient whose reduction dimension is the batch */
    long split_stride;
    void *ws;          /* optional fp32 scratch (ws_bytes).  When given and the launch would leave most CUs idle
                          (tiles x 2 <= CUs: a few frames), the product is cut into K-batches whose partial sums a
                          second kernel adds up and finishes with the epilogue: the launch takes a fraction of one
                          tile's K loop instead of all of it (serving latency).  The fp32 summation order then
                          depends on M; NULL keeps every row's result independent of the batch it sits in. */
    size_t ws_bytes;
    int transposed;    /* 1: both operands lie TRANSPOSED -- A is [rows, M] at row stride lda, W is [rows, N] at row
                          stride ldw, and C[m][n] = sum over rows of A[row][m] W[row][n] (a weight gradient dY^T X
                          straight from the activations as the forward and backward passes left them: no transposed
                          copies).  K = rows per batch (a multiple of 64), batch s reads rows s*K ..; EC_EPI_STORE32,
                          no bias, variant 0, M a multiple of 8. */
    int k_rows;        /* transposed: the rows that exist (<= splits * K); rows past it read as zero */
    /* EC_EPI_STORE16_LN / EC_EPI_GELU16_LN: */
    const float *row_stats;   /* fp32 pairs (rstd, -rstd * mean) of the A rows; row m at row_stats + 2 * m * row_stats_stride.
                                 16-byte aligned, M pairs: at stride 1 the kernel fetches them two at a time through a
                                 descriptor whose range ends at pair M - 1 (nothing past the array is read); M < 2^27 */
    long row_stats_stride;    /* in rows (0 = 1): the class-token rows of a [n, S] statistics array are S apart */
    const float *col_sums;    /* fp32 [N]: sum over k of W[n][k] as rounded to 16 bit */
    /* EC_EPI_RESID_HL, optional (N % 64 == 0): */
    float *row_sums;          /* fp32 [M][N / 64][2]: (sum, sum of squares) of the NEW hi plane over each 64-column group
                                 of every row; ec_row_stats_merge turns them into the row statistics of the GEMM that
                                 follows, so no pass over the hi plane is needed */
    /* Split-precision operands in ONE launch (variant 0; EC_EPI_STORE16 / GELU16 / STORE32 / RESID32 / RESID_HL; no
     * splits, ws, resid or transposed operands).  A value is carried as hi = round16(x), lo = round16(x - hi); with
     * A_lo and / or W_lo given, C = epi(A_lo . W^T + A . W_lo^T + A . W^T + bias): up to three MFMA products over K
     * columns each, the small ones first, into the SAME fp32 accumulators and through ONE epilogue (where the three
     * ec_gemm launches of rounds 1 - 4 read and wrote an fp32 C twice more).  A NULL lo part = that operand IS its
     * 16-bit value: its product is skipped (a product with zeros would add nothing: the same bits). */
    const void *A_lo;         /* [M, K] 16-bit at row stride lda, or NULL */
    const void *W_lo;         /* [N, K] 16-bit at row stride ldw, or NULL */
    /* ... and with them EC_EPI_STORE16 / EC_EPI_GELU16 take `aux` as a SECOND OUTPUT: C16 = hi = round16(v) and
     * aux16 = lo = round16(v - hi) of the epilogue's fp32 value v, [M, N] at stride ldc -- the operand pair of a
     * split-precision consumer (ec_attention_split; A_lo of the next GEMM). */
    /* Lo products on the FP8 matrix path (round 6, ABI 600; appended).  A lo part is ~2^-11 of its hi part and its product
     * needs ~2^-4 relative accuracy to cut the operand-rounding error of the 16-bit product 20-fold: e4m3 operands on
     * v_mfma_scale_f32_16x16x128_f8f6f4 run at twice the f16 rate and are half the bytes.  dtype EC_F16, K % 128 == 0, the
     * conditions of A_lo / W_lo; epilogues STORE16 / GELU16 (with or without the lo output) / STORE32 / RESID_HL.
     *   A_lo8 + W8:  C += dq(A_lo8) . dq(W8)^T   in place of A_lo . W^T   (A_lo must be NULL)
     *   A8 + W_lo8:  C += dq(A8) . dq(W_lo8)^T   in place of A . W_lo^T   (W_lo must be NULL)
     * where dq(X8) = e4m3 value x 2^-x_exp: X8 holds round_e4m3(x . 2^x_exp), OCP e4m3fn, one power-of-two scale per tensor
     * (producers: ec_layernorm_hl8 for activations; weights are quantised when they are packed).  LAYOUT: an e4m3 operand
     * lies at the SAME BYTE ROW PITCH as its 16-bit counterpart -- row m of A_lo8 / A8 starts at byte m * 2 * lda and
     * holds K bytes, row n of W8 / W_lo8 at byte n * 2 * ldw -- so that the staging DMA addresses all parts alike. */
    const void *A_lo8;        /* e4m3 of (a_true - A) . 2^a_lo8_exp, or NULL */
    const void *W8;           /* e4m3 of W . 2^w8_exp */
    const void *A8;           /* e4m3 of A . 2^a8_exp */
    const void *W_lo8;        /* e4m3 of (w_true - W) . 2^w_lo8_exp, or NULL */
    int a_lo8_exp, w8_exp, a8_exp, w_lo8_exp;
    int aux_e4m3;             /* != 0 (EC_EPI_GELU16 with aux and A_lo8): the lo output leaves as e4m3 of lo . 2^aux_exp, one byte per
                                 element in the first N bytes of rows of 2 . ldc bytes -- the A_lo8 operand of the GEMM that follows */
    int aux_exp;
} ec_gemm_args;

EC_API int ec_gemm(const ec_gemm_args *args, ec_stream_t stream);

/* LayerNorm (fp32 statistics, eps inside the sqrt) of fp32 rows into the 16-bit
 * GEMM operand.  row_idx (optional, device int32 [rows]) gathers source rows:
 * used for ln_post on the CLS rows and ln_final on the EOT rows. */
EC_API int ec_layernorm(const float *x, long ldx, const int32_t *row_idx, const float *gamma,
                        const float *beta, int rows, int width, float eps, void *out16, long ldo,
                        int dtype, ec_stream_t stream);

/* (rstd, -rstd * mean) of `rows` 16-bit rows of `width` elements at row stride ldx -> stats fp32 [rows][2]: the
 * LayerNorm statistics (eps inside the sqrt) of the hi plane of the residual stream, what EC_EPI_STORE16_LN /
 * EC_EPI_GELU16_LN multiply and add.  Reads 2 bytes per element where ec_layernorm reads 4 and writes 2. */
EC_API int ec_row_stats(const void *x16, long ldx, int rows, int width, float eps, float *stats, int dtype,
                        ec_stream_t stream);

/* (sum, sum of squares) per 64-column group as EC_EPI_RESID_HL leaves them (ec_gemm_args.row_sums, `groups` = width / 64
 * groups of row r at sums + 2 * r * groups * sums_stride ... in rows: row r at sums + 2 * groups * r) -> stats
 * fp32 [rows][2] = (rstd, -rstd * mean) with variance = E[x^2] - mean^2 in fp32 (the hi values are 16-bit numbers:
 * their sums over <= 2048 columns are exact to ~1e-7; relative error of the variance ~1e-7 (1 + mean^2 / variance)). */
EC_API int ec_row_stats_merge(const float *sums, int rows, int groups, int width, float eps, float *stats,
                              ec_stream_t stream);

/* Split-precision helpers ("precise" towers): a value is carried as two 16-bit numbers
 * hi = round16(x), lo = round16(x - hi), and a product x.w as xh.wh + xh.wl + xl.wh with
 * fp32 accumulation (three ec_gemm launches, EC_EPI_STORE32 then EC_EPI_RESID32), which
 * reproduces fp32 arithmetic to ~1e-6 on the 16-bit MFMA path. */
EC_API int ec_layernorm_split(const float *x, long ldx, const int32_t *row_idx, const float *gamma,
                              const float *beta, int rows, int width, float eps, void *out16,
                              void *out16_lo, long ldo, int dtype, ec_stream_t stream);
/* LayerNorm of rows given as the two planes of the folded chain's residual stream (x_hi in `dtype`, x_lo fp16, both at row
 * stride ldx) into hi / lo parts at row stride ldo: the LayerNorm of the split-operand blocks (precise_blocks). */
EC_API int ec_layernorm_hl(const void *x_hi, const void *x_lo, long ldx, const float *gamma, const float *beta, int rows,
                           int width, float eps, void *out16, void *out16_lo, long ldo, int dtype, ec_stream_t stream);
/* fp32 [n] -> (QuickGELU if gelu) -> hi / lo 16-bit parts */
/* ... with the lo part as the e4m3 operand of ec_gemm_args.A_lo8 (round 6): out_lo8 = round_e4m3((LN(x) - out16) . 2^lo_exp) and,
 * when out_hi8 is given, out_hi8 = round_e4m3(out16 . 2^hi_exp) (ec_gemm_args.A8), one byte per element in the first `width`
 * bytes of rows of 2 . ldo bytes (the byte row pitch of out16); values beyond +-448 . 2^-exp saturate.  f16 planes. */
EC_API int ec_layernorm_hl8(const void *x_hi, const void *x_lo, long ldx, const float *gamma, const float *beta, int rows,
                            int width, float eps, void *out16, void *out_lo8, void *out_hi8, long ldo, int lo_exp, int hi_exp,
                            ec_stream_t stream);
EC_API int ec_split16(const float *x, long n, int gelu, void *hi16, void *lo16, int dtype,
                      ec_stream_t stream);
/* fp32 attention over fp32 qkv [n_seq * S, 3 * width]; output as hi / lo parts [n_seq * S, width] */
EC_API int ec_attention_f32(const float *qkv, void *out_hi, void *out_lo, int n_seq, int S,
                            int width, int heads, int causal, int dtype, ec_stream_t stream);

/* fp32 attention (no mask) over q | k | v given as hi + lo 16-bit parts, two [n_seq * S, 3 * width] tensors as
 * EC_EPI_STORE16 leaves them with ec_gemm_args.aux; output as hi / lo parts [n_seq * S, width].  q_prescaled != 0: the q
 * columns already hold q * log2(e) / sqrt(64) (ec_vit_weights.q_scaled), else a plain q.  The attention of the first
 * split-operand blocks (ec_vit_weights.precise_attn_blocks).  A plain f16 q and a sequence whose K_hi, K_lo, V_hi, V_lo fit a CU's LDS in one pass
 * (4 * roundup32(S) * 128 bytes, plus 2 KB where a lone last query tile is split over the waves, within 160 KiB: S <= 320,
 * S <= 288 for such sequences) or in two passes over the keys (S <= 608, the tiles' state in registers between them): three 16-bit MFMA products
 * per score and per P.V tile (the lo . lo terms left out), ~1e-6 from float64; otherwise scores, softmax and P.V in fp32 on
 * v_mfma_f32_16x16x4_f32. */
EC_API int ec_attention_split(const void *qkv_hi, const void *qkv_lo, void *out_hi, void *out_lo, int n_seq, int S,
                              int width, int heads, int q_prescaled, int dtype, ec_stream_t stream);

/* x[n, 0] = class_embedding, x[n, 1 + p] = patch[n * (seq - 1) + p]; + positional
 * embedding; ln_pre -> fp32 residual stream x [n_img, seq, width]. */
EC_API int ec_vit_embed(const float *patch, const float *cls, const float *pos, const float *gamma,
                        const float *beta, int n_img, int seq, int width, float eps, float *x,
                        ec_stream_t stream);

/* The same, additionally keeping the un-normalised embedding `pre` (fp32 [n_img, seq, width]; NULL = none):
 * ln_pre's backward pass differentiates through it. */
EC_API int ec_vit_embed_train(const float *patch, const float *cls, const float *pos, const float *gamma,
                              const float *beta, int n_img, int seq, int width, float eps, float *x,
                              float *pre, ec_stream_t stream);

/* x[n, s] = token_embedding[tokens[n, s]] + positional_embedding[s]. */
EC_API int ec_text_embed(const int32_t *tokens, const float *table, const float *pos, int n_txt,
                         int ctx, int width, int vocab, float *x, ec_stream_t stream);

/* Multi-head self-attention, head dim 64 (nn.MultiheadAttention of the CLIP
 * blocks).  qkv: 16-bit [n_seq * S, 3 * width] = q | k | v; out: 16-bit
 * [n_seq * S, width].  causal != 0 applies the text tower's mask. */
EC_API int ec_attention(const void *qkv, void *out, int n_seq, int S, int width, int heads,
                        int causal, int dtype, ec_stream_t stream);
/* Same, computing only the first q_rows query rows of every sequence (keys and values still span the
 * whole sequence); out is [n_seq * q_rows, width].  q_rows = 1 is what the LAST block of the vision
 * tower needs: encode_image reads nothing but the class token of its output. */
EC_API int ec_attention_rows(const void *qkv, void *out, int n_seq, int S, int width, int heads,
                             int causal, int q_rows, int dtype, ec_stream_t stream);
/* ec_attention_rows for a qkv buffer whose q columns already hold q * log2(e) / sqrt(64) (the softmax
 * temperature and the base change folded into the q rows of in_proj before their rounding to 16 bit:
 * ec_vit_weights.q_scaled): the kernel's scores are the exponent's arguments as they leave the MFMA.  What the
 * image tower calls; the other entry points multiply a plain q themselves. */
EC_API int ec_attention_scaled_q(const void *qkv, void *out, int n_seq, int S, int width, int heads,
                                 int causal, int q_rows, int dtype, ec_stream_t stream);

/* Training forms (fine-tuning the vision tower, models/clip_cls_ft.py:44-80): the same forward that also
 * keeps, per (sequence, head, query), the log2 of its softmax denominator in the scaled-score domain
 * (lse fp32 [n_seq, heads, S]), and the backward pass torch autograd runs for nn.MultiheadAttention:
 * d_out 16-bit [n_seq * S, width] -> d_qkv 16-bit [n_seq * S, 3 * width] (dq | dk | dv).
 * delta: fp32 scratch [n_seq, heads, S].  No causal mask (the text tower is never trained). */
EC_API int ec_attention_train(const void *qkv, void *out, float *lse, int n_seq, int S, int width,
                              int heads, int dtype, ec_stream_t stream);
EC_API int ec_attention_backward(const void *qkv, const void *out, const float *lse, const void *d_out,
                                 void *d_qkv, float *delta, int n_seq, int S, int width, int heads,
                                 int dtype, ec_stream_t stream);

/* ------------------------------------------------------------------------
 * CLIP towers.  Replace clip_model.encode_image / encode_text as called from
 * models/clip_cls.py:101 / :84 (module structure: un-vendored openai/CLIP
 * clip/model.py VisionTransformer / Transformer / ResidualAttentionBlock).
 * Weight pointers are borrowed device memory; 16-bit tensors are in `dtype`.
 * ------------------------------------------------------------------------ */
typedef struct {
    const float *ln1_g, *ln1_b; /* ln_1 [W] */
    const void *qkv_w;          /* attn.in_proj_weight [3W, W] 16-bit */
    const float *qkv_b;         /* attn.in_proj_bias [3W] */
    const void *out_w;          /* attn.out_proj.weight [W, W] 16-bit */
    const float *out_b;
    const float *ln2_g, *ln2_b; /* ln_2 [W] */
    const void *fc1_w;          /* mlp.c_fc.weight [4W, W] 16-bit */
    const float *fc1_b;
    const void *fc2_w;          /* mlp.c_proj.weight [W, 4W] 16-bit */
    const float *fc2_b;
    /* lo parts (w - round16(w)) of the four matrices; read by precise towers and by the split-operand blocks of
     * ec_vit_weights.precise_blocks (whose qkv_w / qkv_b hold a PLAIN q), else NULL.  NULL in such a block = the matrix
     * is its 16-bit value (ec_vit_weights.weights_exact16). */
    const void *qkv_w_lo, *out_w_lo, *fc1_w_lo, *fc2_w_lo;
    /* LayerNorm folded into the two GEMMs that consume it (ec_vit_weights.ln_folded; EC_EPI_*_LN): the weight
     * with its columns scaled by the LayerNorm gain BEFORE the rounding to 16 bit, W' = W diag(gamma) [N, W]; the
     * fp32 sum of every row of W' as rounded [N]; and the bias with the LayerNorm offset folded in, b + W beta [N]
     * (ln_1 into in_proj, ln_2 into c_fc).  NULL otherwise. */
    const void *qkv_w_ln;
    const float *qkv_cs, *qkv_bf;
    const void *fc1_w_ln;
    const float *fc1_cs, *fc1_bf;
    /* (round 6, ABI 600; appended) e4m3 operands of the split-operand blocks' lo products (ec_vit_weights.lo_fp8; ec_gemm_args.W8 /
     * W_lo8): round_e4m3(w16 . 2^exp) of the PLAIN 16-bit matrices qkv_w / fc1_w / fc2_w and round_e4m3(w_lo . 2^exp) of the lo
     * parts of qkv_w / fc1_w, each [N, K] bytes at the 16-bit byte row pitch (rows of 2 K bytes, the first K used); NULL = not packed
     * (a lo part that is NULL stays NULL). */
    const void *qkv_w8, *fc1_w8, *fc2_w8, *qkv_wlo8, *fc1_wlo8;
    int qkv_w8_exp, fc1_w8_exp, fc2_w8_exp, qkv_wlo8_exp, fc1_wlo8_exp;
} ec_block_weights;

typedef struct {
    int dtype;                  /* EC_F16 / EC_BF16 */
    int image_size, patch, width, layers, heads, out_dim;
    int kpad;                   /* patch-row width: 6 * patch * patch rounded up to a multiple of 64
                                   (EC_PRE_PATCHES16: [hi | lo | 0]) */
    const void *conv_w;         /* visual.conv1.weight [W, 3 p^2] in (c, i, j) order, rounded to 16 bits
                                   and laid out against the patch row: [W, kpad] = [w_hi | w_hi | 0] */
    const float *cls, *pos;     /* class_embedding [W], positional_embedding [S, W] */
    const float *ln_pre_g, *ln_pre_b, *ln_post_g, *ln_post_b;
    const void *proj_w;         /* visual.proj transposed: [out_dim, W] 16-bit */
    const ec_block_weights *blocks; /* host array [layers] */
    int precise;                /* != 0: split-precision arithmetic in the blocks too (3x the GEMM work,
                                   ~fp32 results).  The patch embedding and ln_post @ proj always run
                                   split (hi + lo operands): they are 0.6 % of the flops and would be
                                   half of the logit error otherwise */
    const void *conv_w_lo;      /* [W, roundup64(3 p^2)] = [w - w_hi rounded to 16 bits | 0]; NULL with weights_exact16 when all zero */
    const void *proj_w_lo;      /* [out_dim, W] lo part of proj_w; NULL with weights_exact16 when all zero */
    int full_last_block;        /* encode_image returns ln_post(x[:, 0]) @ proj (openai/CLIP model.py), so
                                   of the last block's output only the class-token rows are ever read.
                                   0 (default): that block computes keys / values for every token but the
                                   attention query, out_proj, ln_2 and the MLP for the class token only --
                                   the same features bit for bit.  != 0: every token, like the reference. */
    int low_latency;            /* != 0: under-filled GEMM launches (batches of a few frames) run K-batched
                                   (ec_gemm_args.ws): a single frame of ViT-L/14 in about half the time, at the
                                   price of results that differ from the large-batch ones in the last fp32 bits.
                                   0 (default): a frame's features do not depend on the batch it is in. */
    int q_scaled;               /* != 0: the q rows of every block's qkv_w / qkv_b (rows 0 .. W-1 of in_proj) were
                                   multiplied by log2(e) / sqrt(64) in fp32 BEFORE the rounding to 16 bit, and the
                                   blocks call ec_attention_scaled_q: the attention kernel then has nothing to
                                   scale -- the product is rounded once, like an unscaled q.  Inference only:
                                   ec_vit_train_forward and precise towers take a plain q (0). */
    int ln_folded;              /* != 0: the blocks' LayerNorms run folded into the GEMMs around them (the *_ln / *_cs /
                                   *_bf fields of every block are set): the residual stream is kept as hi + lo 16-bit
                                   planes, the residual GEMMs update it in that form (EC_EPI_RESID_HL), ec_row_stats
                                   reads the hi plane (2 bytes per element instead of LayerNorm's 4 + 2) and the QKV /
                                   c_fc GEMMs take the raw hi rows and finish LayerNorm in their epilogues
                                   (EC_EPI_STORE16_LN / EC_EPI_GELU16_LN).  The SAME NUMBER of 16-bit roundings as the
                                   plain chain, at different places: the plain chain rounds LN(x), this one rounds x
                                   itself (the hi plane) and normalises afterwards with statistics of those rounded
                                   values, the variance taken as E[x^2] - mean^2 in fp32 -- a form that loses digits
                                   on rows whose mean^2 is much larger than their variance (measured: 2.4e-4 against
                                   2.1e-4 feature error on ViT-L/14; tests/test_outliers_gpu.py holds the
                                   outlier-channel statistics of released checkpoints).  Ignored by precise /
                                   low_latency towers and by the training entry points. */
    int precise_blocks;         /* 0 < precise_blocks < layers with precise == 0, ln_folded != 0, dtype EC_F16: the FIRST
                                   precise_blocks blocks are split-operand blocks on the SAME hi + lo planes of the
                                   residual stream: LayerNorm of both planes into hi + lo parts (ec_layernorm_hl), QKV and
                                   c_fc multiplying both parts (ec_gemm_args.A_lo), all four GEMMs adding the product with
                                   their weight's lo part where it has one (W_lo) -- every product of a GEMM in ONE launch.
                                   Their matrices are the PLAIN ones (qkv_w with an unscaled q, fc1_w: neither the LayerNorm
                                   gain nor the softmax scale folded in, so that a checkpoint stored in 16 bit has no lo
                                   parts and pays one product less per GEMM).  A rounding error made in an early block is
                                   carried through every later one, and the residual-stream operand and the weights are
                                   where most of it is made (tools/rounding_budget.py).  (Rounds 1 - 4 ran
                                   these blocks as the fp32-stream chain of `precise`, three launches per GEMM and fp32
                                   attention everywhere: 2.0 x the step; this form costs 1.3 - 1.5 x.)  0 (default): off. */
    int weights_exact16;        /* != 0: the blocks' 16-bit matrices ARE the weights (a checkpoint stored in 16 bit, as
                                   clip.load() returns one on a GPU): the qkv_w_lo / out_w_lo / fc1_w_lo / fc2_w_lo of a
                                   split-precision block may be NULL, and the x_hi . w_lo product of such a matrix -- a sum
                                   of zeros -- is skipped (two MFMA products per GEMM instead of three, the same bits).
                                   0 (default): a split-precision block without its lo parts is an error. */
    int precise_attn_blocks;    /* (round 5, appended) <= precise_blocks: in the first precise_attn_blocks of the split-operand blocks the QKV
                                   GEMM also writes the lo parts of q | k | v (ec_gemm_args.aux) and attention runs in fp32
                                   on hi + lo (ec_attention_split), its output entering out_proj as hi + lo, and c_fc writes
                                   QuickGELU's output as hi + lo into c_proj: where attention is sharp the 16-bit rounding
                                   of q and k in the FIRST blocks is the largest single error and that of the MLP activation
                                   the next (tools/rounding_budget.py; profiles/r5_tolerance_sweep.txt); behind them the
                                   16-bit attention kernel on a plain q with the scores scaled in fp32. */
    int lo_fp8;                 /* (round 6, appended) != 0 with precise_blocks: the lo products of the split-operand blocks' QKV, c_fc and
                                   c_proj GEMMs run on the FP8 matrix path -- LayerNorm writes its lo part (and, where the weight has
                                   a lo part, a copy of its hi part) as e4m3 (ec_layernorm_hl8), c_fc writes the MLP activation's lo
                                   part as e4m3 (ec_gemm_args.aux_e4m3), and the GEMMs multiply them with the e4m3 weights of
                                   ec_block_weights (*_w8 / *_wlo8) at twice the f16 rate and half the operand bytes: an e4m3 lo
                                   product measures 0.51 - 0.52 of the f16 one.  The lo part is ~2^-11 of its hi part, so e4m3's 2^-4
                                   still removes ~95 % of the 16-bit operand-rounding error (profiles/r6_fp8_model_8_5.txt, r6_parity_seeds.txt).
                                   out_proj (the attention output's lo part) stays 16-bit.  Needs the *_w8 fields of every split-operand block. */
} ec_vit_weights;

typedef struct {
    int dtype;
    int ctx, vocab, width, layers, heads, out_dim;
    const float *token_embedding;   /* [vocab, W] fp32 */
    const float *pos;               /* [ctx, W] */
    const float *ln_final_g, *ln_final_b;
    const void *proj_w;             /* text_projection transposed: [out_dim, W] 16-bit */
    const ec_block_weights *blocks; /* host array [layers] */
    int precise;                    /* != 0: split-precision arithmetic (text features are cached) */
    const void *proj_w_lo;
} ec_text_weights;

/* bytes of scratch needed to push `chunk` images (or texts) through at once */
EC_API size_t ec_vit_workspace_bytes(const ec_vit_weights *w, int chunk);
EC_API size_t ec_text_workspace_bytes(const ec_text_weights *w, int chunk);

/* patches: 16-bit [n_img, G, kpad] with G = (image_size / patch)^2 (what
 * ec_preprocess or ec_patchify writes); feats: fp32 [n_img, out_dim] (out).
 * Images are processed `chunk` at a time through `workspace`. */
EC_API int ec_vit_encode(const ec_vit_weights *w, const void *patches, int n_img, float *feats,
                         void *workspace, size_t workspace_bytes, int chunk, ec_stream_t stream);

/* tokens: int32 [n_txt, ctx]; feats: fp32 [n_txt, out_dim] = ln_final(x)[EOT] @
 * text_projection, EOT = argmax of the token ids (not yet L2-normalised). */
EC_API int ec_text_encode(const ec_text_weights *w, const int32_t *tokens, int n_txt, float *feats,
                          void *workspace, size_t workspace_bytes, int chunk, ec_stream_t stream);


/* ------------------------------------------------------------------------
 * Logits and view aggregation.  Replaces ZSCLIPClassifier.forward
 * (models/clip_cls.py:131-162: logits :148, scatter :151-152,
 * _aggregate_logits :104-121, _aggregate_probs :123-129) and the tail of
 * FSCLIPClassifier.forward (:326-343: F.normalize, mask, logits).
 * ------------------------------------------------------------------------ */
enum { EC_AGG_SUM = 0, EC_AGG_MEAN = 1, EC_AGG_MAX = 2 };

/* text_t:  fp32 [C, K] text features, transposed.  Prepared ONCE per set of text features (they are constant across
 *          batches: cached zero-shot prompts, clip_cls.py:84-93; a learned parameter changes once per optimiser step):
 *          ec_classify_prep_text writes the hi + lo fp16 planes of the K class rows (each scaled by its own power of
 *          two so that the lo part is a normal number) and the scales into text_ws (ec_classify_text_bytes(C, K)
 *          bytes, 256-byte aligned), which ec_classify_v2 reads.
 * feats:   fp32 [n_rows, C] image features (compact over valid views, or full).
 * row_idx: int32 [B, T]: row of feats for view (b, t), or -1 for an invalid view.
 * normalize != 0: L2-normalise each view's features first (F.normalize, eps 1e-12).
 * Outputs fp32: full_logits [B, T, K] (invalid views 0), logits [B, K], probs [B, K].
 * The product runs on the matrix pipe at fp32 accuracy: both operands as hi + lo fp16 parts, three MFMA products in
 * one ec_gemm launch (ec_gemm_args.A_lo / W_lo), the scales taken out again exactly.
 * workspace: ec_classify_v2_workspace_bytes(n_rows, C, K) bytes, 256-byte aligned (per call; no state survives).
 * ec_classify: the name of rounds 1 - 5 (other argument lists); kept as a stub that returns EC_ERR_UNSUPPORTED so that a
 * caller built against an older header gets an error code, not a mis-read argument list. */
EC_API size_t ec_classify_text_bytes(int C, int K);
EC_API int ec_classify_prep_text(const float *text_t, int C, int K, void *text_ws, size_t text_ws_bytes, ec_stream_t stream);
EC_API size_t ec_classify_v2_workspace_bytes(int n_rows, int C, int K);
EC_API int ec_classify_v2(const float *feats, int n_rows, const int32_t *row_idx, const void *text_ws, int B, int T,
                          int C, int K, float logit_scale, int agg, int normalize, float *full_logits,
                          float *logits, float *probs, void *workspace, size_t workspace_bytes, ec_stream_t stream);
EC_API int ec_classify(void);

/* ------------------------------------------------------------------------
 * Few-shot feature adapter.  Replaces TransformerAdapter.forward
 * (models/adapter.py:82-105) + Adapter.residual_add (:22-25) including the
 * scatter of view features into a zero [B, T, C] tensor (models/clip_cls.py:319-321).
 * fp32.  All Linear weights are handed over TRANSPOSED ([in, out] row-major).
 * ------------------------------------------------------------------------ */
typedef struct {
    const float *ln1_g, *ln1_b;    /* norm1 [d_model] */
    const float *qkv_w_t, *qkv_b;  /* self_attn.in_proj_weight^T [d_model, 3 d_model], bias */
    const float *o_w_t, *o_b;      /* self_attn.out_proj.weight^T [d_model, d_model] */
    const float *ln2_g, *ln2_b;    /* norm2 */
    const float *w1_t, *b1;        /* linear1.weight^T [d_model, ffn] */
    const float *w2_t, *b2;        /* linear2.weight^T [ffn, d_model] */
} ec_adapter_layer;

typedef struct {
    int in_dim, d_model, heads, ffn, layers;
    float residual;                /* r of in * r + new * (1 - r) */
    const float *in_w_t, *in_b;    /* in_proj.weight^T [in_dim, d_model] */
    const float *out_w_t, *out_b;  /* out_proj.weight^T [d_model, in_dim] */
    const ec_adapter_layer *layer; /* host array [layers] */
} ec_adapter_weights;

/* feats: fp32 [n_rows, in_dim]; row_idx: int32 [B, T] (row of feats or -1 = padded
 * view, which enters as a zero row and is masked as an attention key);
 * out: fp32 [B, T, in_dim]. */
EC_API int ec_adapter_forward(const ec_adapter_weights *w, const float *feats,
                              const int32_t *row_idx, int B, int T, float *out, ec_stream_t stream);

/* ---- pseudo-label selection (gen_data.py:132-164) -------------------
 * probs: float32 [B, V, K], the classifier's `probs` for the V views of each sample (V = 4 with
 * test-time augmentation, event2img.py:94-112; V = 1 without).  Outputs per sample: the view-mean
 * distribution (optional, [B, K]), its argmax and maximum, and selected = 1 when
 * max > conf_thresh, and (tta_consistent) every view predicts the same class, and (tta_min_prob)
 * the smallest per-view top probability > conf_thresh. */
EC_API int ec_pseudo_label(const float *probs, int B, int V, int K, float conf_thresh,
                           int tta_consistent, int tta_min_prob, float *mean_probs, int32_t *pred,
                           float *max_prob, uint8_t *selected, ec_stream_t stream);

/* ---- few-shot `text-identity` training step ---------------------------
 * One optimisation step of FSCLIPClassifier with adapter_type = 'text-identity'
 * (configs/fsclip/text_adapter/) on cached image features: forward models/clip_cls.py:302-350,
 * loss :164-175, d loss / d text_feats (torch autograd upstream), torch.optim.Adam.
 *   img_feats   fp32 [B, T, D] encoder outputs, NOT normalised (any value on invalid views)
 *   valid       uint8 [B, T];  labels int32 [B];  text_param fp32 [K, D] (the nn.Parameter, raw)
 *   agg         EC_AGG_SUM / EC_AGG_MEAN;  use_probs_loss: 0 = CE on the aggregated logits,
 *               1 = NLL of log(probs + 1e-6) (loss_dict.use_probs_loss)
 *   loss        fp32 [1] (out, batch mean);  grad_text fp32 [K, D] (out)
 *   agg_logits  optional fp32 [B, K] (out): the aggregated logits of the forward pass */
EC_API size_t ec_fs_text_train_workspace_bytes(int B, int T, int D, int K);
EC_API int ec_fs_text_loss_grad(const float *img_feats, const uint8_t *valid, const int32_t *labels,
                                const float *text_param, int B, int T, int D, int K, float logit_scale,
                                int agg, int use_probs_loss, float *loss, float *grad_text,
                                float *agg_logits, void *workspace, size_t workspace_bytes,
                                ec_stream_t stream);
/* torch.optim.Adam (no amsgrad), in place; step counts from 1 */
EC_API int ec_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                        float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                        ec_stream_t stream);

/* 'text-trans' (configs/fsclip/joint_adapter/): the same step with the TransformerAdapter
 * (models/adapter.py:52-110: in_proj -> nn.TransformerEncoder(norm_first, ReLU, key padding mask) ->
 * out_proj -> residual mix) in front of the normalisation, differentiated with respect to every
 * adapter parameter and text_feats.  Parameter tensors are passed in torch's own layouts
 * ([out, in] weights, state-dict names in the comments), gradients come back in a second struct of the
 * same shape.  img_feats must hold ZERO rows for invalid views (clip_cls.py:319-321).
 * dropout_p > 0 applies the four dropouts of nn.TransformerEncoderLayer in train mode (attention
 * weights, after out_proj, inside the MLP, after linear2; p = 0.1 upstream) with masks from a
 * stateless hash of (dropout_seed, site, element) -- the same distribution as torch's, not its random
 * stream; site = 4 * layer + {0, 1, 2, 3} in that order, element = flat index into [B, heads, T, T] /
 * [B*T, d_model] / [B*T, ffn_dim] / [B*T, d_model].  ec_dropout_mask returns the keep mask of a site
 * (tests replay it in the oracle).  dropout_p = 0: the deterministic, eval-mode function. */
typedef struct {
    float *ln1_g, *ln1_b;   /* norm1.weight / .bias [d] */
    float *qkv_w, *qkv_b;   /* self_attn.in_proj_weight [3d, d] / in_proj_bias [3d] */
    float *o_w, *o_b;       /* self_attn.out_proj.weight [d, d] / .bias */
    float *ln2_g, *ln2_b;   /* norm2 */
    float *w1, *b1;         /* linear1.weight [ffn, d] / .bias */
    float *w2, *b2;         /* linear2.weight [d, ffn] / .bias */
} ec_adapter_train_layer;

typedef struct {
    int in_dim, d_model, heads, ffn_dim, layers; /* layers <= 8 */
    float residual;                              /* Adapter.residual (adapter.py:13-20) */
    float *in_w, *in_b;                          /* in_proj.weight [d, in_dim] / .bias */
    float *out_w, *out_b;                        /* out_proj.weight [in_dim, d] / .bias */
    const ec_adapter_train_layer *blocks;        /* host array [layers] */
} ec_adapter_train_params;

EC_API size_t ec_fs_trans_train_workspace_bytes(int B, int T, int D, int K, int d_model, int ffn_dim,
                                                int heads, int layers);
EC_API int ec_fs_trans_loss_grad(const float *img_feats, const uint8_t *valid, const int32_t *labels,
                                 const float *text_param, int B, int T, int D, int K, float logit_scale,
                                 int agg, int use_probs_loss, const ec_adapter_train_params *params,
                                 const ec_adapter_train_params *grads, float dropout_p,
                                 uint64_t dropout_seed, float *loss, float *grad_text,
                                 float *agg_logits, void *workspace, size_t workspace_bytes,
                                 ec_stream_t stream);
EC_API int ec_dropout_mask(uint64_t seed, uint32_t site, int64_t n, float p, uint8_t *mask,
                           ec_stream_t stream);

/* C[M, N] (row stride ldc) = alpha * sum_k A(m, k) B(k, n) + beta * C with A(m, k) = A[m * sam + k * sak],
 * B(k, n) = B[k * sbk + n * sbn]: fp32, any layout, for the SMALL products around the towers (LoRA factors
 * up @ down and their gradients, models/lora.py:50-52,138-150; the projection head). */
EC_API int ec_sgemm(const float *A, long sam, long sak, const float *B, long sbk, long sbn, int M, int N, int K,
                    float alpha, float beta, float *C, long ldc, ec_stream_t stream);

/* ---- fine-tuning the vision tower ----------------------------------------------
 * One training step of FTCLIPClassifier (models/clip_cls_ft.py) runs CLIP's visual encoder under autograd:
 * forward :180-183 -> F.normalize / logits / aggregation :196-243 -> calc_train_loss :245-256 -> backward
 * into whatever _build_clip (:44-80) left trainable (all of model.visual, sub-sets, or LoRA factors that act
 * through merged weights W + up @ down, models/lora.py:138-150) -> Adam (method.py:152-186).  Here:
 *   ec_vit_train_forward   encode_image keeping what the backward pass needs (per block: both residual
 *                          inputs, both LayerNorm outputs, q | k | v, the attention output and its log-sum-exp,
 *                          the MLP pre-activation and its QuickGELU: 36 bytes per token and channel);
 *   ec_ft_loss_grad        the classifier head: loss and d loss / d image features, d loss / d text_feats;
 *   ec_vit_train_backward  d features -> gradients of every visual parameter asked for, in the layouts of
 *                          the state dict (fp32; weight gradients dY^T X straight from the row-major
 *                          activations: ec_gemm_args.transposed); unused gradients: NULL pointers; LoRA factor
 *                          gradients through ec_vit_lora (rank-r products from the activations);
 *   ec_pack_weight16_batched  fp32 master weights -> the 16-bit operand copies the kernels read;
 *   ec_grad_unscale_check  the gradient-scaler step of mixed-precision training (`--fp16`, train.py:121).
 * Arithmetic: 16-bit MFMA operands (activations, weights, activation gradients), fp32 accumulation, fp32
 * residual-stream gradients, LayerNorm and softmax in fp32 -- what torch.cuda.amp does for the reference.
 * Gradients of the loss times `scale` flow through when d_feats is scaled (f16's range); they are linear
 * in d_feats.  Every token of the last block is computed (full_last_block semantics). */
typedef struct {
    const void *qkv_wt;  /* attn.in_proj_weight TRANSPOSED [W, 3W] 16-bit (ec_pack_weight16 hi_t) */
    const void *out_wt;  /* attn.out_proj.weight^T [W, W] */
    const void *fc1_wt;  /* mlp.c_fc.weight^T [W, 4W] */
    const void *fc2_wt;  /* mlp.c_proj.weight^T [4W, W] */
} ec_block_weights_t;

typedef struct {
    const ec_block_weights_t *blocks; /* host array [layers] */
    const float *proj;                /* visual.proj [W, out_dim] fp32 */
} ec_vit_train_weights;

/* Gradient outputs, fp32, overwritten (not accumulated), state-dict layouts.  NULL = not wanted: the work
 * that only serves that gradient is skipped (a weight gradient is one extra GEMM, biases and LayerNorm
 * terms are reductions). */
typedef struct {
    float *ln1_g, *ln1_b, *qkv_w /* [3W, W] */, *qkv_b, *out_w /* [W, W] */, *out_b;
    float *ln2_g, *ln2_b, *fc1_w /* [4W, W] */, *fc1_b, *fc2_w /* [W, 4W] */, *fc2_b;
} ec_block_grads;

typedef struct {
    float *conv_w;                    /* visual.conv1.weight [W, 3, p, p] */
    float *cls, *pos;                 /* class_embedding [W], positional_embedding [S, W] */
    float *ln_pre_g, *ln_pre_b, *ln_post_g, *ln_post_b;
    float *proj;                      /* visual.proj [W, out_dim] */
    const ec_block_grads *blocks;     /* host array [layers] */
} ec_vit_grads;

/* LoRA factors of the attention projections (models/lora.py), differentiated straight from the activations:
 * y = x (W + up down)^T gives d up = dy^T (x down^T) and d down = (dy up)^T x, two rank-r products per
 * projection -- the W x W gradient of the merged weight is never formed.  Index 0 .. 3 = q, k, v, out_proj;
 * a projection without factors has d_up NULL.  down16: the factor [r, W] as 16 bit, rows padded with zeros to a
 * multiple of 16; up16_t: up TRANSPOSED [r, W], padded the same way (ec_pack_weight16_batched writes both);
 * d_up [W, r] and d_down [r, W]: fp32 out, the factors' own layouts. */
typedef struct {
    const void *down16[4], *up16_t[4];
    float *d_up[4], *d_down[4];
} ec_block_lora;
typedef struct {
    int rank;                        /* r <= 64 */
    const ec_block_lora *blocks;     /* host array [layers] */
} ec_vit_lora;

/* The scalars that change from step to step (learning rates and Adam's bias corrections, the loss scale and
 * its inverse) can be read from a DEVICE array of EC_STEP_COUNT floats instead of the by-value arguments: a
 * step recorded into a hipGraph is then replayed with new values by overwriting that array.  NULL = by value. */
enum { EC_STEP_LR0 = 0, EC_STEP_LR1 = 1, EC_STEP_BC1 = 2, EC_STEP_BC2_SQRT = 3, EC_STEP_GRAD_SCALE = 4,
       EC_STEP_INV_SCALE = 5, EC_STEP_COUNT = 8 };

/* bytes of workspace for n_img images: the saved activations + the backward pass's scratch.  The forward
 * call fills it, the backward call must get the same buffer back untouched. */
EC_API size_t ec_vit_train_workspace_bytes(const ec_vit_weights *w, int n_img);
EC_API int ec_vit_train_forward(const ec_vit_weights *w, const void *patches, int n_img, float *feats,
                                void *workspace, size_t workspace_bytes, ec_stream_t stream);
/* d_feats fp32 [n_img, out_dim].  patches: the forward call's input (conv1's gradient reads it). */
EC_API int ec_vit_train_backward(const ec_vit_weights *w, const ec_vit_train_weights *wt, const void *patches,
                                 int n_img, const float *d_feats, const ec_vit_grads *grads,
                                 const ec_vit_lora *lora /* NULL = none */, void *workspace,
                                 size_t workspace_bytes, ec_stream_t stream);

/* The same pass in pieces: stage 0 = the head (ln_post, proj), stages 1 .. layers = blocks layers - 1 .. 0, stage
 * layers + 1 = the embedding; [stage_begin, stage_end) of them run, in order, and the state in between lives in
 * the workspace.  For data-parallel training the caller runs a few blocks, starts the all-reduce of their
 * (finished) slice of the gradient buffer on the collective's own stream, and carries on with the next blocks:
 * the bucketed, overlapped gradient exchange of DistributedDataParallel (train.py --ddp). */
EC_API int ec_vit_train_backward_stages(const ec_vit_weights *w, const ec_vit_train_weights *wt, const void *patches,
                                        int n_img, const float *d_feats, const ec_vit_grads *grads,
                                        const ec_vit_lora *lora, int stage_begin, int stage_end, void *workspace,
                                        size_t workspace_bytes, ec_stream_t stream);

/* fp32 [rows, cols] -> any of: hi = round16(w) [rows, cols]; lo = round16(w - hi); hi_t = hi transposed
 * [cols, rows] (NULL outputs are skipped), for a list of same-shape matrices in one launch (`items`: DEVICE array):
 * what the optimiser moved is repacked once per step. */
typedef struct {
    const float *w;
    void *hi, *lo, *hi_t;
} ec_pack_item;
EC_API int ec_pack_weight16_batched(const ec_pack_item *items, int n_items, int rows, int cols, int dtype,
                                    ec_stream_t stream);

/* LayerNorm backward (fp32): x rows at stride ldx (the forward input), dy rows at stride ldy.
 * dx rows at stride ldo: dx = (accumulate ? dx : 0) + dLN; d_gamma / d_beta [width] (NULL = skip; they need
 * `partials`, fp32 scratch of ec_layernorm_backward_partials(rows, width) floats). */
EC_API size_t ec_layernorm_backward_partials(int rows, int width);
EC_API int ec_layernorm_backward(const float *x, long ldx, const float *dy, long ldy, const float *gamma, int rows,
                                 int width, float eps, float *dx, long ldo, int accumulate, float *d_gamma,
                                 float *d_beta, float *partials, ec_stream_t stream);

/* The classifier head of FTCLIPClassifier in train mode (clip_cls_ft.py:196-256; identity adapter):
 * ec_fs_text_loss_grad plus d loss / d img_feats scaled by grad_scale (1 = none).
 * row_idx NULL: img_feats / grad_img are fp32 [B, T, D] (zero gradient rows for invalid views).
 * row_idx int32 [B, T] (row of view (b, t), -1 = invalid view): img_feats / grad_img are compact [Nv, D] over
 * the valid views, exactly what the encoder produced and what ec_vit_train_backward takes -- the scatter
 * of :208-209 and its gather in the backward pass never materialise.
 * grad_text may be NULL (fixed text features).  Workspace: ec_fs_text_train_workspace_bytes(B, T, D, K) +
 * max(B * T, K) * D * 4 bytes. */
EC_API int ec_ft_loss_grad(const float *img_feats, const int32_t *row_idx, const uint8_t *valid,
                           const int32_t *labels, const float *text_param, int B, int T, int D, int K,
                           float logit_scale, int agg, int use_probs_loss, float grad_scale,
                           const float *step_scalars /* NULL, or [EC_STEP_GRAD_SCALE] replaces grad_scale */, float *loss,
                           float *grad_text, float *grad_img, float *agg_logits, void *workspace,
                           size_t workspace_bytes, ec_stream_t stream);

/* LoRA factors (models/lora.py), every injected projection of the tower in one launch.  Per projection:
 * out = base + up @ down (base / out fp32 [rows, cols], up [rows, r], down [r, cols], r <= 64; lora.py:50-52,
 * :138-150), and the chain rule from the merged weight's gradient: d_up = dW down^T, d_down = up^T dW.
 * `items` is a DEVICE array; all items share rows, cols, r (CLIP's q / k / v / out projections are W x W). */
typedef struct {
    const float *base, *up, *down;
    float *out;
    const float *dW;
    float *d_up, *d_down;
} ec_lora_item;
EC_API int ec_lora_merge_batched(const ec_lora_item *items, int n_items, int rows, int cols, int r, ec_stream_t stream);
EC_API size_t ec_lora_grad_scratch_floats(int n_items, int rows, int cols, int r);
EC_API int ec_lora_grad_batched(const ec_lora_item *items, int n_items, int rows, int cols, int r, float *scratch,
                                ec_stream_t stream);

/* ec_adam_step over a list of tensors in one launch (`items`: DEVICE array; group selects lr0 / lr1: the
 * classifier's own parameters vs model.visual, method.py:166-178); max_n = the largest item's n.
 * skip_flag (device int32 or NULL): when it reads non-zero at execution time nothing is updated -- the
 * found_inf of ec_grad_unscale_check, so that a dropped mixed-precision step needs no host round trip. */
typedef struct {
    float *param;
    const float *grad;
    float *exp_avg, *exp_avg_sq;
    int64_t n;
    int group;
} ec_adam_item;
EC_API int ec_adam_step_multi(const ec_adam_item *items, int n_items, int64_t max_n, float lr0, float lr1, float beta1,
                              float beta2, float eps, float weight_decay, int step, const int32_t *skip_flag,
                              const float *step_scalars /* NULL, or lr0 / lr1 / 1 - beta1^t / sqrt(1 - beta2^t) */,
                              ec_stream_t stream);

/* grad *= inv_scale in place; *found_inf (device int32, caller zeroes it once per step) is set when any
 * element is not finite -- torch.cuda.amp.GradScaler.unscale_. */
EC_API int ec_grad_unscale_check(float *grad, int64_t n, float inv_scale, int32_t *found_inf,
                                 const float *step_scalars /* NULL, or [EC_STEP_INV_SCALE] replaces inv_scale */,
                                 ec_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EVENTCLIP_HIP_H */
