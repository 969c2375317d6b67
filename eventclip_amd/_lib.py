"""ctypes binding of libeventclip_hip.so (the C ABI declared in include/eventclip_hip.h).

There is no CPU fallback: if the library is missing or no MI355X is visible the
callers raise.  torch is used only to own device memory and streams.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# EVENTCLIP_HIP_LIB: tools/ point it at libeventclip_hip_diag.so (python -m eventclip_amd.build --diag),
# the build whose ec_gemm also has the stamp / timeline / timing-experiment variants
LIB_PATH = os.environ.get('EVENTCLIP_HIP_LIB') or os.path.join(_HERE, 'libeventclip_hip.so')
DIAG_LIB_PATH = os.path.join(_HERE, 'libeventclip_hip_diag.so')

EC_F16, EC_BF16 = 0, 1

c_void_p, c_int, c_long, c_double, c_float = (ctypes.c_void_p, ctypes.c_int, ctypes.c_long,
                                              ctypes.c_double, ctypes.c_float)


class EcFrameStats(ctypes.Structure):
    _fields_ = [('sum', ctypes.c_uint64), ('sumsq', ctypes.c_uint64), ('nnz', ctypes.c_uint32),
                ('max_kept', ctypes.c_uint32), ('dropped', ctypes.c_uint32),
                ('ambiguous', ctypes.c_uint32), ('thr', ctypes.c_double)]


class EcEventsParams(ctypes.Structure):
    _fields_ = [('H', c_int), ('W', c_int), ('thresh', c_double), ('count_non_zero', c_int),
                ('background_mask', c_int), ('red', ctypes.c_uint8 * 3),
                ('blue', ctypes.c_uint8 * 3), ('max_frame_events', c_int), ('flip_x', c_int),
                ('negate_p', c_int), ('sort_workspace', c_void_p),
                ('sort_workspace_bytes', ctypes.c_size_t), ('float32_stage', c_int),
                ('total_events', ctypes.c_int64)]


class EcAdapterLayer(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ('ln1_g', 'ln1_b', 'qkv_w_t', 'qkv_b', 'o_w_t', 'o_b',
                                        'ln2_g', 'ln2_b', 'w1_t', 'b1', 'w2_t', 'b2')]


class EcAdapterWeights(ctypes.Structure):
    _fields_ = [('in_dim', c_int), ('d_model', c_int), ('heads', c_int), ('ffn', c_int),
                ('layers', c_int), ('residual', c_float), ('in_w_t', c_void_p), ('in_b', c_void_p),
                ('out_w_t', c_void_p), ('out_b', c_void_p),
                ('layer', ctypes.POINTER(EcAdapterLayer))]


class EcAugOp(ctypes.Structure):
    _fields_ = [('kind', c_int), ('alpha', c_float), ('param', c_double), ('m', c_double * 6)]


(EC_AUG_IDENTITY, EC_AUG_AFFINE, EC_AUG_ROT180, EC_AUG_ROT90, EC_AUG_ROT270, EC_AUG_BRIGHTNESS,
 EC_AUG_COLOR, EC_AUG_CONTRAST, EC_AUG_SHARPNESS, EC_AUG_POSTERIZE, EC_AUG_SOLARIZE,
 EC_AUG_AUTOCONTRAST, EC_AUG_EQUALIZE) = range(13)


class EcProfileEntry(ctypes.Structure):
    _fields_ = [('name', ctypes.c_char * 64), ('launches', c_long), ('total_ms', c_double),
                ('flops', c_double), ('bytes', c_double)]


class EcGemmArgs(ctypes.Structure):
    _fields_ = [('M', c_int), ('N', c_int), ('K', c_int), ('dtype', c_int), ('epilogue', c_int),
                ('variant', c_int), ('A', c_void_p), ('lda', c_long), ('W', c_void_p),
                ('bias', c_void_p), ('C', c_void_p), ('ldc', c_long), ('diag', c_void_p),
                ('ldw', c_long), ('resid', c_void_p), ('aux', c_void_p), ('splits', c_int),
                ('split_stride', c_long), ('ws', c_void_p), ('ws_bytes', ctypes.c_size_t),
                ('transposed', c_int), ('k_rows', c_int), ('row_stats', c_void_p), ('row_stats_stride', c_long),
                ('col_sums', c_void_p), ('row_sums', c_void_p), ('A_lo', c_void_p), ('W_lo', c_void_p),
                ('A_lo8', c_void_p), ('W8', c_void_p), ('A8', c_void_p), ('W_lo8', c_void_p),
                ('a_lo8_exp', c_int), ('w8_exp', c_int), ('a8_exp', c_int), ('w_lo8_exp', c_int),
                ('aux_e4m3', c_int), ('aux_exp', c_int)]


EC_EPI_STORE16, EC_EPI_GELU16, EC_EPI_RESID32, EC_EPI_STORE32 = 0, 1, 2, 3
EC_EPI_GELU16_SAVE, EC_EPI_GELU_BWD16 = 4, 5
EC_EPI_RESID_HL, EC_EPI_STORE16_LN, EC_EPI_GELU16_LN = 6, 7, 8
EC_STEP_LR0, EC_STEP_LR1, EC_STEP_BC1, EC_STEP_BC2_SQRT, EC_STEP_GRAD_SCALE, EC_STEP_INV_SCALE, EC_STEP_COUNT = 0, 1, 2, 3, 4, 5, 8
EC_PRE_CHW_F32, EC_PRE_PATCHES16, EC_PRE_HWC_U8 = 0, 1, 2
EC_AGG_SUM, EC_AGG_MEAN, EC_AGG_MAX = 0, 1, 2
EC_OK, EC_ERR_INVALID, EC_ERR_HIP, EC_ERR_UNSUPPORTED, EC_ERR_WORKSPACE = 0, -1, -2, -3, -4


class EcBlockWeights(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in (
        'ln1_g', 'ln1_b', 'qkv_w', 'qkv_b', 'out_w', 'out_b', 'ln2_g', 'ln2_b', 'fc1_w', 'fc1_b',
        'fc2_w', 'fc2_b', 'qkv_w_lo', 'out_w_lo', 'fc1_w_lo', 'fc2_w_lo', 'qkv_w_ln', 'qkv_cs', 'qkv_bf', 'fc1_w_ln',
        'fc1_cs', 'fc1_bf', 'qkv_w8', 'fc1_w8', 'fc2_w8', 'qkv_wlo8', 'fc1_wlo8')] + [
        (n, c_int) for n in ('qkv_w8_exp', 'fc1_w8_exp', 'fc2_w8_exp', 'qkv_wlo8_exp', 'fc1_wlo8_exp')]


class EcAdapterTrainLayer(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ('ln1_g', 'ln1_b', 'qkv_w', 'qkv_b', 'o_w', 'o_b', 'ln2_g', 'ln2_b',
                                        'w1', 'b1', 'w2', 'b2')]


class EcAdapterTrainParams(ctypes.Structure):
    _fields_ = [('in_dim', c_int), ('d_model', c_int), ('heads', c_int), ('ffn_dim', c_int), ('layers', c_int),
                ('residual', ctypes.c_float), ('in_w', c_void_p), ('in_b', c_void_p), ('out_w', c_void_p),
                ('out_b', c_void_p), ('blocks', ctypes.POINTER(EcAdapterTrainLayer))]


class EcVitWeights(ctypes.Structure):
    _fields_ = [('dtype', c_int), ('image_size', c_int), ('patch', c_int), ('width', c_int),
                ('layers', c_int), ('heads', c_int), ('out_dim', c_int), ('kpad', c_int),
                ('conv_w', c_void_p), ('cls', c_void_p), ('pos', c_void_p),
                ('ln_pre_g', c_void_p), ('ln_pre_b', c_void_p), ('ln_post_g', c_void_p),
                ('ln_post_b', c_void_p), ('proj_w', c_void_p),
                ('blocks', ctypes.POINTER(EcBlockWeights)), ('precise', c_int),
                ('conv_w_lo', c_void_p), ('proj_w_lo', c_void_p), ('full_last_block', c_int),
                ('low_latency', c_int), ('q_scaled', c_int), ('ln_folded', c_int), ('precise_blocks', c_int), ('weights_exact16', c_int),
                ('precise_attn_blocks', c_int), ('lo_fp8', c_int)]


class EcTextWeights(ctypes.Structure):
    _fields_ = [('dtype', c_int), ('ctx', c_int), ('vocab', c_int), ('width', c_int),
                ('layers', c_int), ('heads', c_int), ('out_dim', c_int),
                ('token_embedding', c_void_p), ('pos', c_void_p), ('ln_final_g', c_void_p),
                ('ln_final_b', c_void_p), ('proj_w', c_void_p),
                ('blocks', ctypes.POINTER(EcBlockWeights)), ('precise', c_int),
                ('proj_w_lo', c_void_p)]


class EcLoraItem(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ('base', 'up', 'down', 'out', 'dW', 'd_up', 'd_down')]


class EcPackItem(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ('w', 'hi', 'lo', 'hi_t')]


class EcAdamItem(ctypes.Structure):
    _fields_ = [('param', c_void_p), ('grad', c_void_p), ('exp_avg', c_void_p), ('exp_avg_sq', c_void_p),
                ('n', ctypes.c_int64), ('group', c_int)]


class EcBlockWeightsT(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ('qkv_wt', 'out_wt', 'fc1_wt', 'fc2_wt')]


class EcVitTrainWeights(ctypes.Structure):
    _fields_ = [('blocks', ctypes.POINTER(EcBlockWeightsT)), ('proj', c_void_p)]


BLOCK_GRAD_FIELDS = ('ln1_g', 'ln1_b', 'qkv_w', 'qkv_b', 'out_w', 'out_b', 'ln2_g', 'ln2_b', 'fc1_w', 'fc1_b',
                     'fc2_w', 'fc2_b')


class EcBlockGrads(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in BLOCK_GRAD_FIELDS]


class EcVitGrads(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ('conv_w', 'cls', 'pos', 'ln_pre_g', 'ln_pre_b', 'ln_post_g', 'ln_post_b',
                                        'proj')] + [('blocks', ctypes.POINTER(EcBlockGrads))]


class EcBlockLora(ctypes.Structure):
    _fields_ = [('down16', c_void_p * 4), ('up16_t', c_void_p * 4), ('d_up', c_void_p * 4), ('d_down', c_void_p * 4)]


class EcVitLora(ctypes.Structure):
    _fields_ = [('rank', c_int), ('blocks', ctypes.POINTER(EcBlockLora))]


# name -> (restype, argtypes); kept in one table so tests can check that every
# symbol of the header is exported.
SIGNATURES = {
    'ec_last_error': (ctypes.c_char_p, []),
    'ec_version': (c_int, []),
    'ec_abi_check': (c_int, [c_int] + [ctypes.c_size_t] * 6),
    'ec_device_info': (c_int, [ctypes.POINTER(c_int), ctypes.c_char_p, c_int]),
    'ec_profile_begin': (c_int, []),
    'ec_profile_end': (c_int, [ctypes.POINTER(EcProfileEntry), c_int, ctypes.POINTER(c_int)]),
    'ec_events_to_frames': (c_int, [c_void_p, c_void_p, c_int, ctypes.POINTER(EcEventsParams),
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'ec_events_sort_workspace_bytes': (ctypes.c_size_t, [ctypes.POINTER(EcEventsParams)]),
    'ec_center_events': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'ec_augment_events': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'ec_pack_events': (c_int, [c_void_p, ctypes.c_int64, c_void_p, c_void_p, c_void_p]),
    'ec_events_to_frames_packed': (c_int, [c_void_p, c_void_p, c_int, ctypes.POINTER(EcEventsParams),
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'ec_center_events_packed': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'ec_pseudo_label': (c_int, [c_void_p, c_int, c_int, c_int, ctypes.c_float, c_int, c_int, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_void_p]),
    'ec_fs_text_train_workspace_bytes': (ctypes.c_size_t, [c_int, c_int, c_int, c_int]),
    'ec_fs_text_loss_grad': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                     ctypes.c_float, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                     ctypes.c_size_t, c_void_p]),
    'ec_fs_trans_train_workspace_bytes': (ctypes.c_size_t, [c_int] * 8),
    'ec_fs_trans_loss_grad': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                      ctypes.c_float, c_int, c_int, ctypes.POINTER(EcAdapterTrainParams),
                                      ctypes.POINTER(EcAdapterTrainParams), ctypes.c_float, ctypes.c_uint64,
                                      c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_size_t, c_void_p]),
    'ec_dropout_mask': (c_int, [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int64, ctypes.c_float, c_void_p,
                                c_void_p]),
    'ec_adam_step': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_int64, ctypes.c_float,
                             ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, c_int, c_void_p]),
    'ec_gemm': (c_int, [ctypes.POINTER(EcGemmArgs), c_void_p]),
    'ec_preprocess_plan_bytes': (ctypes.c_size_t, [c_int, c_int, c_int]),
    'ec_preprocess_plan': (c_int, [c_int, c_int, c_int, c_void_p, ctypes.c_size_t]),
    'ec_preprocess': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                              c_int, c_void_p]),
    'ec_patchify': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    'ec_layernorm': (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float,
                             c_void_p, c_long, c_int, c_void_p]),
    'ec_layernorm_split': (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                   c_float, c_void_p, c_void_p, c_long, c_int, c_void_p]),
    'ec_split16': (c_int, [c_void_p, c_long, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    'ec_attention_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                 c_int, c_void_p]),
    'ec_attention_split': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'ec_vit_embed': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                             c_float, c_void_p, c_void_p]),
    'ec_text_embed': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                              c_void_p]),
    'ec_attention': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                             c_void_p]),
    'ec_row_stats_merge': (c_int, [c_void_p, c_int, c_int, c_int, ctypes.c_float, c_void_p, c_void_p]),
    'ec_row_stats': (c_int, [c_void_p, c_long, c_int, c_int, ctypes.c_float, c_void_p, c_int, c_void_p]),
    'ec_layernorm_hl': (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p,
                                c_long, c_int, c_void_p]),
    'ec_layernorm_hl8': (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p,
                                 c_void_p, c_long, c_int, c_int, c_void_p]),
    'ec_attention_scaled_q': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                      c_void_p]),
    'ec_attention_rows': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                  c_void_p]),
    'ec_vit_workspace_bytes': (ctypes.c_size_t, [ctypes.POINTER(EcVitWeights), c_int]),
    'ec_text_workspace_bytes': (ctypes.c_size_t, [ctypes.POINTER(EcTextWeights), c_int]),
    'ec_vit_encode': (c_int, [ctypes.POINTER(EcVitWeights), c_void_p, c_int, c_void_p, c_void_p,
                              ctypes.c_size_t, c_int, c_void_p]),
    'ec_text_encode': (c_int, [ctypes.POINTER(EcTextWeights), c_void_p, c_int, c_void_p, c_void_p,
                               ctypes.c_size_t, c_int, c_void_p]),
    'ec_adapter_forward': (c_int, [ctypes.POINTER(EcAdapterWeights), c_void_p, c_void_p, c_int, c_int,
                                   c_void_p, c_void_p]),
    'ec_randaugment_workspace_bytes': (ctypes.c_size_t, [c_int, c_int, c_int, c_int]),
    'ec_randaugment': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p,
                               c_void_p, ctypes.c_size_t, c_void_p]),
    'ec_classify_text_bytes': (ctypes.c_size_t, [c_int, c_int]),
    'ec_classify_prep_text': (c_int, [c_void_p, c_int, c_int, c_void_p, ctypes.c_size_t, c_void_p]),
    'ec_classify_v2_workspace_bytes': (ctypes.c_size_t, [c_int, c_int, c_int]),
    'ec_classify_v2': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                               c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_size_t, c_void_p]),
    'ec_classify': (c_int, []),          # stub of the rounds 1 - 5 name: EC_ERR_UNSUPPORTED
    'ec_attention_train': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'ec_attention_backward': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                      c_int, c_int, c_int, c_void_p]),
    'ec_vit_embed_train': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                   c_float, c_void_p, c_void_p, c_void_p]),
    'ec_sgemm': (c_int, [c_void_p, c_long, c_long, c_void_p, c_long, c_long, c_int, c_int, c_int, c_float,
                         c_float, c_void_p, c_long, c_void_p]),
    'ec_vit_train_workspace_bytes': (ctypes.c_size_t, [ctypes.POINTER(EcVitWeights), c_int]),
    'ec_vit_train_forward': (c_int, [ctypes.POINTER(EcVitWeights), c_void_p, c_int, c_void_p, c_void_p,
                                     ctypes.c_size_t, c_void_p]),
    'ec_vit_train_backward': (c_int, [ctypes.POINTER(EcVitWeights), ctypes.POINTER(EcVitTrainWeights), c_void_p,
                                      c_int, c_void_p, ctypes.POINTER(EcVitGrads), ctypes.POINTER(EcVitLora), c_void_p,
                                      ctypes.c_size_t, c_void_p]),
    'ec_vit_train_backward_stages': (c_int, [ctypes.POINTER(EcVitWeights), ctypes.POINTER(EcVitTrainWeights), c_void_p,
                                             c_int, c_void_p, ctypes.POINTER(EcVitGrads), ctypes.POINTER(EcVitLora), c_int,
                                             c_int, c_void_p, ctypes.c_size_t, c_void_p]),
    'ec_pack_weight16_batched': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'ec_layernorm_backward_partials': (ctypes.c_size_t, [c_int, c_int]),
    'ec_layernorm_backward': (c_int, [c_void_p, c_long, c_void_p, c_long, c_void_p, c_int, c_int, c_float,
                                      c_void_p, c_long, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'ec_ft_loss_grad': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                ctypes.c_size_t, c_void_p]),
    'ec_grad_unscale_check': (c_int, [c_void_p, ctypes.c_int64, c_float, c_void_p, c_void_p, c_void_p]),
    'ec_lora_merge_batched': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'ec_lora_grad_scratch_floats': (ctypes.c_size_t, [c_int, c_int, c_int, c_int]),
    'ec_lora_grad_batched': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'ec_adam_step_multi': (c_int, [c_void_p, c_int, ctypes.c_int64, c_float, c_float, c_float, c_float, c_float,
                                   c_float, c_int, c_void_p, c_void_p, c_void_p]),

}

_lib = None


ABI_VERSION = 600      # EC_ABI_VERSION of include/eventclip_hip.h these bindings mirror


class HipLibraryError(RuntimeError):
    pass


def lib():
    """Load the library once.  Raises HipLibraryError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                f'{LIB_PATH} is missing: build it with `python -m eventclip_amd.build` '
                '(hipcc --offload-arch=gfx950).  eventclip_amd has no CPU fallback.')
        # One HIP runtime per process: torch bundles its own libamdhip64 (soname
        # libamdhip64.so.7).  Load it first so that this library's NEEDED entry
        # binds to the same runtime instead of pulling a second copy from
        # /opt/rocm, whose streams and device state torch would not share.
        import torch
        bundled = os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so')
        if os.path.exists(bundled):
            ctypes.CDLL(bundled, mode=ctypes.RTLD_GLOBAL)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        # the ctypes mirrors above against the library's own struct definitions (include/eventclip_hip.h, EC_ABI_CHECK)
        rc = handle.ec_abi_check(ABI_VERSION, *(ctypes.sizeof(t) for t in (EcGemmArgs, EcBlockWeights, EcVitWeights,
                                                                           EcTextWeights, EcEventsParams, EcAdapterWeights)))
        if rc != 0:
            raise HipLibraryError(f'{LIB_PATH}: {handle.ec_last_error().decode()} (eventclip_amd/_lib.py binds ABI '
                                  f'{ABI_VERSION}; rebuild with `python -m eventclip_amd.build`)')
        _lib = handle
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().ec_last_error()
        raise RuntimeError(f'{what or "libeventclip_hip"} failed ({rc}): '
                           f'{msg.decode() if msg else ""}')


def require_gpu():
    """The device every op runs on; raises when there is none."""
    import torch
    if not torch.cuda.is_available():
        raise HipLibraryError('eventclip_amd needs an MI355X (gfx950) device: '
                              'torch.cuda.is_available() is False and there is no CPU fallback.')
    return torch.device('cuda', torch.cuda.current_device())


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def profile_begin():
    check(lib().ec_profile_begin(), 'ec_profile_begin')


def profile_end():
    """-> list of dicts (name, launches, total_ms, flops, bytes), one per kernel symbol."""
    arr = (EcProfileEntry * 32)()
    n = c_int(0)
    check(lib().ec_profile_end(arr, 32, ctypes.byref(n)), 'ec_profile_end')
    return [dict(name=arr[i].name.decode(), launches=int(arr[i].launches),
                 total_ms=float(arr[i].total_ms), flops=float(arr[i].flops),
                 bytes=float(arr[i].bytes)) for i in range(n.value)]
