# -*- coding:utf-8 -*-
"""ctypes binding of libdt_hip.so (the C-ABI declared in include/dt_hip.h).

This is the stub a DeepTables maintainer would add next to deeptables/models/layers.py to call
the MI355X kernels.  `import torch` MUST come first: the library's
NEEDED libamdhip64.so.7 then binds to the HIP runtime torch already loaded, so
`tensor.data_ptr()` and `torch.cuda.current_stream().cuda_stream` are valid in our launches.

There is no CPU fallback: `lib()` raises if the shared object is missing.
"""
import ctypes
import os

import torch  # noqa: F401  (must be imported before the CDLL below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libdt_hip.so')

_c_int = ctypes.c_int
_c_i64 = ctypes.c_int64
_c_f32 = ctypes.c_float
_ptr = ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/dt_hip.h one to one
SIGNATURES = {
    'dt_version': (_c_int, []),
    'dt_last_error': (ctypes.c_char_p, []),
    'dt_build_arch': (ctypes.c_char_p, []),
    'dt_source_hash': (ctypes.c_char_p, []),
    'dt_graph_upload': (_c_int, [_ptr, _ptr]),
    'dt_step_trace': (_c_int, [_c_int]),
    'dt_step_trace_read': (_c_int, [_ptr, _c_int, _ptr]),
    'dt_embedding_fwd': (_c_int, [_ptr, _c_int, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _ptr, _ptr,
                                  _ptr, _ptr]),
    'dt_embedding_bwd_dense': (_c_int, [_ptr, _ptr, _c_int, _c_int, _ptr, _ptr]),
    'dt_fm_fwd': (_c_int, [_ptr, _c_int, _c_int, _c_int, _ptr, _ptr]),
    'dt_fm_bwd': (_c_int, [_ptr, _ptr, _c_int, _c_int, _c_int, _ptr, _ptr]),
    'dt_embed_fm_linear_fwd': (_c_int, [_ptr, _c_int, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int,
                                        _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    'dt_embed_fm_linear_bwd': (_c_int, [_ptr, _ptr, _ptr, _c_int, _ptr, _ptr, _c_int, _c_int, _c_int,
                                        _ptr, _ptr]),
    'dt_bn_workspace_bytes': (_c_i64, [_c_int, _c_int]),
    'dt_bn_train_fwd': (_c_int, [_ptr, _c_int, _c_int, _ptr, _ptr, _c_f32, _c_f32, _ptr, _ptr, _ptr,
                                 _ptr, _ptr, _ptr, _ptr]),
    'dt_bn_infer_fwd': (_c_int, [_ptr, _c_int, _c_int, _ptr, _ptr, _c_f32, _ptr, _ptr, _ptr, _ptr]),
    'dt_bn_train_bwd': (_c_int, [_ptr, _ptr, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr,
                                 _ptr]),
    'dt_cross_workspace_bytes': (_c_i64, [_c_int, _c_int, _c_int]),
    'dt_cross_fwd': (_c_int, [_ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr]),
    'dt_cross_bwd': (_c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr,
                              _ptr, _ptr]),
    'dt_inner_product_fwd': (_c_int, [_ptr, _c_int, _c_int, _c_int, _ptr, _ptr]),
    'dt_inner_product_bwd': (_c_int, [_ptr, _ptr, _c_int, _c_int, _c_int, _ptr, _ptr]),
    'dt_outer_product_fwd': (_c_int, [_ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr]),
    'dt_outer_product_bwd': (_c_int, [_ptr, _ptr, _c_int, _ptr, _c_int, _c_int, _c_int, _ptr, _ptr,
                                      _ptr]),
    'dt_cin_layer_fwd': (_c_int, [_ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_int,
                                  _c_int, _c_i64, _c_i64, _ptr, _ptr]),
    'dt_cin_layer_bwd': (_c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_int,
                                  _c_int, _c_i64, _c_i64, _ptr, _ptr, _ptr, _ptr, _ptr]),
    'dt_cin_bwd_workspace_bytes': (_c_i64, [_c_int] * 5),
    'dt_cin_pool': (_c_int, [_ptr, _c_i64, _c_int, _c_int, _c_int, _ptr, _ptr]),
    'dt_cin_pool_bwd': (_c_int, [_ptr, _ptr, _c_i64, _c_int, _c_int, _c_int, _ptr, _ptr]),
    'dt_cin_layer_bwd_ws': (_c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_int,
                                     _c_int, _c_i64, _c_i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    'dt_cin_bf16_workspace_bytes': (_c_i64, [_c_int, _c_int, _c_int]),
    'dt_cin_layer_fwd_bf16': (_c_int, [_ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_int,
                                       _c_int, _c_i64, _c_i64, _ptr, _ptr, _ptr]),
    'dt_cin_layer_bwd_bf16': (_c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_int,
                                       _c_int, _c_i64, _c_i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    'dt_cin_bf16x3_workspace_bytes': (_c_i64, [_c_int, _c_int, _c_int]),
    'dt_cin_layer_fwd_bf16x3': (_c_int, [_ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_int,
                                         _c_int, _c_i64, _c_i64, _ptr, _ptr, _ptr]),
    'dt_cin_layer_bwd_bf16x3': (_c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_int,
                                         _c_int, _c_i64, _c_i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    'dt_mha_core_fwd': (_c_int, [_ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f32, ctypes.c_uint32, _ptr,
                                 _ptr, _ptr]),
    'dt_mha_core_bwd': (_c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                 _c_f32, ctypes.c_uint32, _ptr, _ptr, _ptr, _ptr]),
    'dt_autoint_supported': (_c_int, [_c_int, _c_int, _c_int]),
    'dt_autoint_dropout_hash': (ctypes.c_uint32, [ctypes.c_uint32] * 5),
    'dt_autoint_fwd': (_c_int, [_ptr] * 9 + [_c_i64, _c_int, _c_int, _c_int, _c_f32, ctypes.c_uint32] + [_ptr] * 6 + [_c_int, _ptr]),
    'dt_autoint_bwd': (_c_int, [_ptr] * 11 + [_c_i64, _c_int, _c_int, _c_int, _c_f32, ctypes.c_uint32] + [_ptr] * 14 + [_c_int, _ptr]),
    'dt_autoint_fwd_bn': (_c_int, [_ptr] * 9 + [_c_i64, _c_int, _c_int, _c_int, _c_f32, ctypes.c_uint32, _ptr, _ptr, _c_f32, _c_f32]
                          + [_ptr] * 12 + [_c_int, _ptr]),
    'dt_autoint_fwd_bn_workspace_bytes': (_c_i64, [_c_i64, _c_int]),
    'dt_autoint_bwd_w': (_c_int, [_ptr] * 11 + [_c_i64, _c_int, _c_int, _c_int, _c_f32, ctypes.c_uint32] + [_ptr] * 19 + [_c_int, _ptr]),
    'dt_autoint_head_workspace_bytes': (_c_i64, [_c_i64, _c_int]),
    'dt_autoint_head_fwd': (_c_int, [_ptr] * 7 + [_c_i64, _c_int, _c_int, _ptr, _ptr]),
    'dt_autoint_head_bwd': (_c_int, [_ptr] * 7 + [_c_i64, _c_int, _c_int] + [_ptr] * 5),
    'dt_autoint_bwd_workspace_bytes': (_c_i64, [_c_i64, _c_int]),
    'dt_bn_train_bwd_stats': (_c_int, [_ptr, _ptr, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    'dt_adam_state_init': (_c_int, [_ptr, _c_f32, _c_f32, _c_f32, _c_int, _ptr]),
    'dt_adam_advance': (_c_int, [_ptr, _c_f32, _c_f32, _c_f32, _ptr]),
    'dt_adam_dense_step': (_c_int, [_ptr, _ptr, _ptr, _ptr, _c_i64, _c_f32, _c_f32, _c_f32, _c_f32,
                                    _ptr, _c_int, _c_f32, _ptr]),
    'dt_adam_multi_step': (_c_int, [_c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _c_f32, _c_f32, _c_f32, _c_f32, _ptr, _c_int,
                                    _c_f32, _ptr]),
    'dt_adam_rows_slots': (_c_i64, [_c_i64]),
    'dt_adam_rows_step': (_c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _c_i64, _c_int, _c_int, _ptr, _c_i64, _ptr, _c_f32,
                                   _c_f32, _c_f32, _c_f32, _ptr, _ptr, _ptr, _ptr, _ptr, _c_i64, _c_int, _c_f32,
                                   _ptr]),
    'dt_adam_rows_step_seg': (_c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _c_i64, _c_int, _c_int, _ptr, _c_i64, _ptr, _c_f32,
                                   _c_f32, _c_f32, _c_f32, _ptr, _ptr, _ptr, _ptr, _ptr, _c_i64, _c_int, _c_f32,
                                   _ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _ptr]),
    'dt_rows_merge_segments': (_c_int, [_ptr, _ptr, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _ptr]),
    'dt_rows_compact': (_c_int, [_ptr, _ptr, _c_i64, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_f32, _c_i64,
                                 _ptr, _ptr, _ptr, _ptr]),
    'dt_bce_logits': (_c_int, [_ptr, _ptr, _c_i64, _ptr, _ptr, _ptr]),
    'dt_sgd_dense_step': (_c_int, [_ptr, _ptr, _c_i64, _c_f32, _ptr]),
    'dt_sgd_rows_step': (_c_int, [_ptr, _ptr, _ptr, _c_i64, _c_int, _c_f32, _ptr]),
    'dt_dense_supported': (_c_int, [_c_int] * 3),
    'dt_dense_workspace_bytes': (_c_i64, [_c_int] * 3),
    'dt_dense_fwd': (_c_int, [_ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr]),
    'dt_dense_bwd': (_c_int, [_ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr]),
    'dt_afm_fwd': (_c_int, [_ptr] * 4 + [_c_int] * 5 + [_ptr] * 3),
    'dt_afm_bwd': (_c_int, [_ptr] * 6 + [_c_int] * 5 + [_ptr] * 5),
    'dt_bilinear_fwd': (_c_int, [_ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr]),
    'dt_bilinear_bwd': (_c_int, [_ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr]),
    'dt_field_pool_fwd': (_c_int, [_ptr, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr]),
    'dt_field_pool_bwd': (_c_int, [_ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr]),
    'dt_field_scale_fwd': (_c_int, [_ptr, _ptr, _c_int, _c_int, _c_int, _ptr, _ptr]),
    'dt_field_scale_bwd': (_c_int, [_ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr]),
    'dt_deepfm_supported': (_c_int, [_c_int] * 6),
    'dt_deepfm_workspace_bytes': (_c_i64, [_c_int] * 4),
    'dt_deepfm_accum_floats': (_c_i64, [_c_int] * 3),
    'dt_deepfm_stamps_offset_floats': (_c_i64, [_c_int] * 4),
    'dt_deepfm_accum_offsets': (_c_int, [_c_int, _c_int, _c_int, _ptr]),
    'dt_deepfm_train_step': (_c_int, [_ptr, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int,
                                      _ptr, _ptr, _ptr, _ptr, _ptr, _c_f32, _c_f32, _ptr, _ptr, _ptr, _ptr, _ptr,
                                      _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _c_i64, _c_f32, _c_int, _c_int,
                                      _c_f32, _ptr, _c_f32, _ptr, _ptr]),
    'dt_deepfm_train_step_adam': (_c_int, [_ptr, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int,
                                           _ptr, _ptr, _ptr, _ptr, _ptr, _c_f32, _c_f32, _ptr, _ptr, _ptr, _ptr, _ptr,
                                           _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _c_i64, _c_int,
                                           _c_f32, _ptr, _c_f32, _ptr, _ptr, _ptr, _c_int, _ptr, _c_f32, _c_f32, _c_f32, _c_f32,
                                           _ptr, _ptr, _ptr, _c_i64, _c_f32, _ptr, _ptr, _ptr, _ptr]),
    'dt_deepfm_step_chains': (_c_int, [_c_int] * 5),
    'dt_deepfm_dropout_hash': (ctypes.c_uint32, [ctypes.c_uint32] * 3),
    'dt_feed_gather': (_c_int, [_ptr, _c_i64, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr]),
    'dt_embedding_gather_owned': (_c_int, [_ptr, _c_int, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_int,
                                           _c_int, _ptr, _ptr, _ptr, _ptr]),
    'dt_deepfm_preelect': (_c_int, [_ptr, _c_int, _ptr, _ptr, _c_int, _c_int, _ptr, _ptr, _c_i64, _ptr]),
    'dt_deepfm_dedupe_slots': (_c_i64, [_c_int, _c_int]),
    'dt_deepfm_dedupe_bytes': (_c_i64, [_c_int, _c_int]),
    'dt_deepfm_dedupe_segments': (_c_int, [_c_int, _c_int, _ptr]),
    'dt_deepfm_dedupe_overflow_offset': (_c_i64, [_c_int, _c_int]),
    'dt_dcn_supported': (_c_int, [_c_int] * 7),
    'dt_dcn_workspace_bytes': (_c_i64, [_c_int] * 5),
    'dt_dcn_stamps_offset_floats': (_c_i64, [_c_int] * 5),
    'dt_dcn_accum_floats': (_c_i64, [_c_int] * 4),
    'dt_dcn_accum_offsets': (_c_int, [_c_int, _c_int, _c_int, _c_int, _ptr]),
    'dt_dcn_train_step': (_c_int, [_ptr, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int,
                                   _ptr, _ptr, _c_int, _ptr, _ptr, _ptr, _ptr, _c_f32, _c_f32,
                                   _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr,
                                   _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _c_i64, _c_int, _c_f32, _ptr, _c_f32, _ptr, _ptr]),
    'dt_dcn_train_step_adam': (_c_int, [_ptr, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int,
                                        _ptr, _ptr, _c_int, _ptr, _ptr, _ptr, _ptr, _c_f32, _c_f32,
                                        _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr,
                                        _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _c_i64, _c_int, _c_f32, _ptr, _c_f32, _ptr,
                                        _ptr, _ptr, _c_int, _ptr, _c_f32, _c_f32, _c_f32, _c_f32,
                                        _ptr, _ptr, _ptr, _c_i64, _c_f32, _ptr, _ptr, _ptr, _ptr]),
}

DT_IDX_F32, DT_IDX_I32 = 0, 1
DT_AI_F32, DT_AI_BF16, DT_AI_BF16X2 = 0, 1, 2
DT_STEP_LOSS_MSE = 0x10
DT_STEP_SKIP_FINISH, DT_STEP_FINISH_ONLY = 0x20, 0x40
DT_STEP_TOWER_X3 = 0x80
DT_STEP_PREELECTED = 0x100
DT_STEP_TOWER_BF16 = 0x200
DT_STEP_STAMPS = 0x400
DT_STEP_PREPARED = 0x800
DT_FEED_CURSOR_WORDS = 528          # 16 (1 + 32 ticket groups), csrc/embedding.hip kFeedGroups
DT_ACT_LINEAR, DT_ACT_RELU = 0, 1
# keras.activations names the CIN / AFM kernels fuse (include/dt_hip.h DT_ACT_*)
ACT_CODES = {None: 0, 'linear': 0, 'relu': 1, 'sigmoid': 2, 'tanh': 3, 'elu': 4, 'selu': 5, 'softplus': 6, 'softsign': 7,
             'exponential': 8}


def act_code(name, what):
    if name not in ACT_CODES:
        raise ValueError(f'{what} activation {name!r}: the HIP kernels fuse {sorted(k for k in ACT_CODES if k)}')
    return ACT_CODES[name]
DT_OP_KERNEL = {'mat': 0, 'vec': 1, 'num': 2}

_lib = None


class DtHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle.  Raises if the HIP library was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                f'(hipcc --offload-arch=gfx950). deeptables_amd has no CPU fallback for the layers hot path.')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().dt_last_error().decode('utf-8', 'replace')
        raise DtHipError(f'{what} failed with code {rc}: {msg}')


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise DtHipError('deeptables_amd kernels run on the GPU only (got a CPU tensor); '
                             'there is deliberately no CPU fallback for the hot path.')
