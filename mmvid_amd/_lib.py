"""ctypes binding of libmmvid_hip.so (declared in include/mmvid_hip.h).

The product path has NO fallback: importing works anywhere (so host logic is testable on CPU), but the
first kernel call without the library or without a GPU raises.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_uint8, c_uint64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MMVID_LIB') or os.path.join(HERE, 'libmmvid_hip.so')  # (MMVID_LIB: another build of the same ABI, for same-box A/Bs)

P = c_void_p
I, I64, F, U64 = c_int, c_int64, c_float, c_uint64


class TowerLayer(Structure):
    _fields_ = [(n, P) for n in (
        'ln1_w', 'ln1_b', 'ln2_w', 'ln2_b', 'in_w', 'in_b', 'out_w', 'out_b', 'fc_w', 'fc_b', 'pj_w', 'pj_b',
        'g_ln1_w', 'g_ln1_b', 'g_ln2_w', 'g_ln2_b', 'g_in_w', 'g_in_b', 'g_out_w', 'g_out_b', 'g_fc_w', 'g_fc_b',
        'g_pj_w', 'g_pj_b')]


class TowerCfg(Structure):
    _fields_ = [('B', I), ('L', I), ('E', I), ('H', I), ('F', I), ('layers', I), ('mask_mode', I), ('r0', I),
                ('c0', I), ('r1', I), ('c1', I), ('ln_eps', F)]


class DecodeToken(Structure):
    """mmvid_decode_token_t (include/mmvid_hip.h)."""
    _fields_ = [('tok', c_void_p), ('table', c_void_p), ('table_rows', c_int64), ('pos_rows', c_void_p), ('pos_off', c_int32),
                ('record_pos0', c_int32), ('record', c_void_p), ('record_ld', c_int64), ('lnf_w', c_void_p), ('lnf_b', c_void_p),
                ('head_w', c_void_p), ('head_b', c_void_p), ('E', c_void_p), ('e_step_stride', c_int64), ('tok_offset', c_int64),
                ('logits_out', c_void_p), ('V', c_int32), ('e_pos0', c_int32), ('lnf_eps', c_float), ('temperature', c_float)]


class VqganOp(Structure):
    _fields_ = [('op', c_int32), ('mode', c_int32), ('N', c_int32), ('H', c_int32), ('W', c_int32), ('C', c_int32),
                ('Cout', c_int32), ('flags', c_int32), ('in0', I64), ('in1', I64), ('in2', I64), ('out_bf16', I64),
                ('out_f32', I64), ('scratch', I64), ('w', P), ('b', P), ('ext_in', P), ('ext_out', P), ('eps', F),
                ('pad', c_int32)]


# name -> argtypes (all return int unless noted)
SIGNATURES = {
    'mmvid_vq_sqnorm': [P, I, I, P, P],
    'mmvid_vq_argmin_l2': [P, P, P, I64, I, I, P, P, P],
    'mmvid_gather_rows': [P, I64, P, I64, I, P, P, P],
    'mmvid_gemm_bf16': [I, I, I, I, I, P, I64, P, I64, I, I64, I64, I64, I, F, P, P, I64, P, P, I64, I, I, P, P, I64, P, P],
    'mmvid_gemm_bf16_dw': [I64, I, I, P, I64, P, I64, I, P, P, I, P],
    'mmvid_gemm_dw_pick_splitk': [I64, I, I],
    'mmvid_gemm_bf16_dw_grouped': [I64, I, I, P, I64, I64, P, I64, I64, I, P, I, P],
    'mmvid_gemm_bf16_dw_multi': [I64, I, P, I, I, P],
    'mmvid_layernorm_fwd': [P, I64, I64, I, P, P, F, P, P, I64, P, P, P],
    'mmvid_layernorm_bwd': [P, I64, P, I64, P, P, P, I64, I, P, I64, I, P, P, P, P, P],
    'mmvid_layernorm_bwd_ws': [P, I64, P, I64, P, P, P, I64, I, P, I64, I, P, P, P, P, P, I64, P],
    'mmvid_layernorm_bwd_partial': [P, I, I64, P, I64, P, P, P, I64, I, P, I64, I, P, I, I, I, P, I64, P, P],
    'mmvid_layernorm_bwd_reduce_multi': [I, P, I, I, P],
    'mmvid_layernorm_bwd_ex': [P, I, I64, P, I64, P, P, P, I64, I, P, I64, I, P, P, P, P, P, I64, P],
    'mmvid_groupnorm_swish_nhwc': [P, I, I, I64, I, P, P, F, I, P, I, P, P, P],
    'mmvid_attention_fwd': [P, I64, I, I, I, I, F, I, I, I, I, I, P, I64, P, P],
    'mmvid_attention_bwd': [P, I64, P, I64, P, I64, P, P, I, I, I, I, F, I, I, I, I, I, P, I64, P],
    'mmvid_attention_bwd_bias': [P, I64, P, I64, P, I64, P, P, I, I, I, I, F, I, I, I, I, I, P, I64, P, P],
    'mmvid_assemble_sequence': [POINTER(P), POINTER(I64), I, P, P, P, I64, I, I, P, P],
    'mmvid_assemble_sequence_bwd': [POINTER(P), POINTER(I64), I, P, P, P, I64, I, I, P, I, P],
    'mmvid_cross_entropy_fwd': [P, I64, P, P, I64, I, P, P, P],
    'mmvid_cross_entropy_bwd': [P, I64, P, P, P, P, I64, I, P, I64, P],
    'mmvid_colsum_bf16': [P, I64, I64, I, P, P],
    'mmvid_grad_sqnorm': [P, I64, P, P],
    'mmvid_adam_step': [P, P, P, P, P, I64, F, F, F, F, F, I, P, F, P, F, P],
    'mmvid_cast_f32_to_bf16': [P, P, I64, P],
    'mmvid_tower_workspace': [POINTER(TowerCfg), POINTER(I64), POINTER(I64)],
    'mmvid_tower_forward': [POINTER(TowerCfg), POINTER(TowerLayer), P, P, P, P, P],
    'mmvid_tower_backward': [POINTER(TowerCfg), POINTER(TowerLayer), P, P, P, P],
    'mmvid_tower_prefill': [POINTER(TowerCfg), POINTER(TowerLayer), P, P, P, I, P, P],
    'mmvid_tower_decode': [POINTER(TowerCfg), POINTER(TowerLayer), P, P, P, I, P, I, P, P],
    'mmvid_tower_decode_fused': [POINTER(TowerCfg), POINTER(TowerLayer), P, P, P, I, P, I, P, P],
    'mmvid_tower_decode_persistent': [POINTER(TowerCfg), POINTER(TowerLayer), P, P, P, I, P, I, I, P, P],
    'mmvid_artv_token_step_persistent': [POINTER(TowerCfg), POINTER(TowerLayer), POINTER(DecodeToken), P, P, I, P, P, P],
    'mmvid_tower_decode_fused_slice': [POINTER(TowerCfg), POINTER(TowerLayer), P, P, P, I, I, P, I, I, P, P],
    'mmvid_gemv_rows': [P, I64, I, I, P, P, F, P, P, I, I, P, I64, I, I, P, I64, P],
    'mmvid_decode_embed': [P, P, I64, P, P, I, I, I, P, P],
    'mmvid_decode_embed_record': [P, P, I64, P, P, I, I, I, P, P, I64, I, P],
    'mmvid_kv_store': [P, I64, I, I, I, P, I, I, P, P],
    'mmvid_attention_decode': [P, I64, P, I, I, I, I, P, I, F, P, I64, P],
    'mmvid_conv2d_nhwc': [I, P, I, I, I, I, P, P, I, P, P, I, P, P, P, P],
    'mmvid_conv2d_nhwc_splitk': [I, P, I, I, I, I, P, P, I, P, P, I, P, P, P, I, P, P],
    'mmvid_conv3x3_strip_supported': [I, I, I, I],
    'mmvid_conv3x3_strip_nhwc': [P, I, I, I, I, P, P, I, P, P, P, P, P, P],
    'mmvid_image_to_nhwc8': [P, I, I, I, P, P],
    'mmvid_nhwc_to_nchw_f32': [P, I, I, I, I, I, P, P],
    'mmvid_spatial_attention': [P, P, P, I, I, I, F, P, P, P],
    'mmvid_spatial_attention_ld': [P, P, P, I64, I, I, I, F, P, P, P],
    'mmvid_conv2d_nhwc_split3': [I, P, I, I, I, I, P, P, I, P, I, P, P, I, P, P],
    'mmvid_conv3x3_strip_nhwc_split3': [P, I, I, I, I, P, P, I, P, P, P, P, P],
    'mmvid_conv3x3_strip_nhwc_f16': [P, I, I, I, I, P, P, I, P, P, P, P, P],
    'mmvid_split_f32_bf16x2': [P, I64, P, P],
    'mmvid_image_to_nhwc8_split': [P, I, I, I, P, P],
    'mmvid_groupnorm_swish_nhwc_split': [P, I, I64, I, P, P, F, I, P, I, P, P],
    'mmvid_groupnorm_swish_nhwc_f16out': [P, I, I64, I, P, P, F, I, P, I, P, P],
    'mmvid_decode_trace': [P],
    'mmvid_decode_persistent_trace': [P],
    'mmvid_vqgan_run': [POINTER(VqganOp), I, P, P],
    'mmvid_sample_race': [P, I64, P, P, F, F, I64, I, I64, P, P, P],
    'mmvid_sample_race_at': [P, I64, P, P, I, I64, P, F, F, I64, I, I64, P, P, P],
    'mmvid_mp_select_keep': [P, P, P, I, I, I, I, P, P],
    'mmvid_mp_build_input': [P, P, I64, P, P, P, I, I, I, I, I, I64, P, P],
    'mmvid_mp_update': [P, P, P, P, P, I, I, I, I, I, P, P, P, P, P, P, P, P, P],
    'mmvid_head_bce_fwd': [P, I64, P, I, I, P, P, F, P, P, P, P, P, I, F, P, P, P, P, P],
    'mmvid_head_bce_bwd': [P, I64, P, I, I, P, P, P, P, P, P, P, P, P, I, F, P, P, I64, P, P, P, P, P],
    'mmvid_bert_build_ids': [P, P, P, P, P, P, I, I, I, I, I64, I64, I, I, P, P, P, P, P],
    'mmvid_grad_sqnorm_det': [P, I64, P, P, P],
    'mmvid_adam_step_lr': [P, P, P, P, P, I64, F, P, F, F, F, F, I, P, F, P, F, P],
    'mmvid_lr_schedule': [P, I, F, F, I, I, P, P],
    'mmvid_grad_sqnorm_rows': [P, I64, P, P, P, I64, I64, I, P],
    'mmvid_adam_step_rows': [P, P, P, P, P, I64, F, P, F, F, F, F, I, P, F, P, F, P, I64, I64, I, P],
    'mmvid_counter_add': [P, F, P],
    'mmvid_msm_masks': [U64, P, I, I, I, P, F, F, F, P, P, P, P],
    'mmvid_msm_masks_inject': [P, P, I, I, I, P, P, P],
    'mmvid_vid_warp': [U64, P, P, I, I, I, I, I, P, P, I, P, P],
    'mmvid_vid_warp_new_frames': [U64, P, P, I, I, I, I, I, P, P, I, P, P],
    'mmvid_vid_warp_tokens': [P, P, P, I, I, I, P, P],
    'mmvid_erase_tokens_choice': [U64, P, I, P, P, P, I, I, I, I, I64, P, P],
    'mmvid_random_erase_tokens': [U64, P, I, I, I, F, F, F, F, F, I, I64, P, P],
    'mmvid_visual_color_jitter': [U64, P, P, I, I, I, I, I, F, I, P, P],
    'mmvid_gemm_f32': [I, I, I, I, P, I64, P, I64, I, I64, I64, I64, F, P, P, P, I64, P],
    'mmvid_conv2d_nhwc_f32': [I, P, I, I, I, I, P, P, I, P, I, P, P],
    'mmvid_image_to_nhwc4_f32': [P, I, I, I, P, P],
    'mmvid_groupnorm_swish_nhwc_f32': [P, I, I64, I, P, P, F, I, P, P, P],
    'mmvid_spatial_attention_f32': [P, P, P, I, I, I, F, P, P, P],
    'mmvid_probe': [I, P, P, P],
    'mmvid_prof_begin': [I],
    'mmvid_prof_enable': [I],
    'mmvid_graph_stats': [P],
    'mmvid_set_option': [c_char_p, I],
    'mmvid_device_faults': [P, I],
    'mmvid_pos_table_fwd': [P, I, I, I, P, P],
    'mmvid_pos_table_bwd': [P, I, I, P, P],
    'mmvid_lincomb3': [P, P, P, F, F, F, P, P],
    'mmvid_scale3': [P, F, F, F, P, P, P, P],
    'mmvid_prof_end': [P, P, P, P, I],
    'mmvid_rows_pack': [P, I64, I, P, I, P, P, P],
    'mmvid_rows_merge': [P, I64, I, P, P, I, P],
}
OTHER = {'mmvid_last_error': ([], c_char_p), 'mmvid_abi_version': ([], I), 'mmvid_device_count': ([], I),
         'mmvid_warp_params_bytes': ([], I), 'mmvid_gemm_dw_multi_fill': ([I, P, I], ctypes.c_double),
         'mmvid_tower_decode_persistent_supported': ([POINTER(TowerCfg), I], I),
         'mmvid_tower_decode_persistent_workspace_bytes': ([I], I64)}

class DwKind(ctypes.Structure):
    """mmvid_dw_kind_t (include/mmvid_hip.h)."""
    _fields_ = [('N', ctypes.c_int32), ('K', ctypes.c_int32), ('dY', ctypes.c_void_p), ('ldy', ctypes.c_int64), ('strideY', ctypes.c_int64),
                ('X', ctypes.c_void_p), ('ldx', ctypes.c_int64), ('strideX', ctypes.c_int64), ('dW_list', ctypes.c_void_p)]


class LnReduce(ctypes.Structure):
    """mmvid_ln_reduce_t (include/mmvid_hip.h)."""
    _fields_ = [('partial', ctypes.c_void_p), ('dw', ctypes.c_void_p), ('db', ctypes.c_void_p), ('dx_colsum', ctypes.c_void_p)]


class PosSegment(ctypes.Structure):
    """mmvid_pos_segment_t (include/mmvid_hip.h)."""
    _fields_ = [('w', ctypes.c_void_p * 3), ('gw', ctypes.c_void_p * 3), ('dst0', ctypes.c_int32), ('rows', ctypes.c_int32),
                ('naxes', ctypes.c_int32), ('src0', ctypes.c_int32), ('d', ctypes.c_int32 * 3), ('pad', ctypes.c_int32)]


ABI_VERSION = 3  # include/mmvid_hip.h: mmvid_abi_version()
_lib = None


class MMVIDError(RuntimeError):
    pass


def load():
    """Load the shared library (no GPU needed for loading).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        # torch's ROCm wheel bundles its own libamdhip64; it must be in the process first so that our library
        # binds to the SAME HIP runtime that owns torch's device pointers and streams (loading ours first gives
        # two runtimes and "no ROCm-capable device" at the first launch).
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise MMVIDError(f'{LIB_PATH} is missing: run `python -m mmvid_amd.build` (hipcc, gfx950). '
                             'There is no CPU/PyTorch fallback for the kernels.')
        lib = ctypes.CDLL(LIB_PATH)
        # the version is checked BEFORE any other symbol is bound: a stale build must say "rebuild", not fail on a missing entry point
        ver = getattr(lib, 'mmvid_abi_version', None)
        have = ver() if ver is not None else None
        if have != ABI_VERSION:
            raise MMVIDError(f'{LIB_PATH} has ABI version {have}, this package needs {ABI_VERSION}: '
                             'rebuild with `python -m mmvid_amd.build --force`')
        for name, args in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = args, I
        for name, (args, res) in OTHER.items():
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = args, res
        _lib = lib
    return _lib


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise MMVIDError(f'{name} failed (rc={rc}): {lib.mmvid_last_error().decode()}')


FAULT_NAMES = ('embedding id outside its table', 'cross-entropy target outside [0, V)', 'reserved', 'reserved')


def device_faults(reset=True):
    """Counts of bad indices the kernels met since the last reset (they read row 0 / class 0 instead of faulting).
    Synchronises the device."""
    arr = (ctypes.c_int64 * 4)()
    call('mmvid_device_faults', arr, int(reset))
    return list(arr)


def check_device_faults():
    """Raise if any kernel met an out-of-range index since the last check (the reference's nn.Embedding / cross_entropy
    would have hit a device-side assert).  Call at a point where a device sync is acceptable: between steps, after a bench."""
    counts = device_faults(reset=True)
    bad = [f'{n} x {FAULT_NAMES[i]}' for i, n in enumerate(counts) if n]
    if bad:
        raise MMVIDError('kernels met out-of-range indices: ' + '; '.join(bad))
