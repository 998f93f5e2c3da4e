"""ctypes binding of libaudiogan_hip.so (C ABI declared in include/audiogan_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C
audiogan_amd/csrc``.  There is NO fallback: if the shared object is missing or
does not export a symbol, importing this module raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libaudiogan_hip.so')

AG_OK, AG_ERR_ARG, AG_ERR_LAUNCH, AG_ERR_UNSUPPORTED = 0, -1, -2, -3
ACT_NONE, ACT_LEAKY, ACT_TANH, ACT_LEAKY_GATE = 0, 1, 2, 3
OPT_RMSPROP, OPT_ADAM = 0, 1
FLAG_NAN, FLAG_BIG = 1, 2

vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class WnDesc(C.Structure):
    _fields_ = [('v', vp), ('g', vp), ('w', vp), ('wpa', vp), ('wpb', vp), ('inv_norm', vp),
                ('rows', i32), ('cols', i32), ('d1', i32), ('K', i32), ('stride', i32), ('pad_', i32)]


class WnBwdDesc(C.Structure):
    _fields_ = [('v', vp), ('g', vp), ('dw', vp), ('dv', vp), ('dg', vp), ('rows', i32), ('cols', i32),
                ('accumulate', i32), ('pad_', i32)]


class ConvArgs(C.Structure):
    _fields_ = [('x', vp), ('wp', vp), ('bias', vp), ('res', vp), ('y', vp), ('lens_i64', vp),
                ('x_bs', i64), ('x_cs', i64), ('y_bs', i64), ('y_cs', i64), ('res_bs', i64),
                ('res_cs', i64), ('B', i32), ('C', i32), ('Lin', i32), ('O', i32), ('Lout', i32),
                ('K', i32), ('stride', i32), ('pad', i32), ('mode', i32), ('act', i32),
                ('slope', f32), ('accumulate', i32), ('wp_pad', i32)]


class OptDesc(C.Structure):
    _fields_ = [('p', vp), ('grad', vp), ('s1', vp), ('s2', vp), ('n', i64)]


# name -> (restype, argtypes); must list EVERY symbol of include/audiogan_hip.h
SIGNATURES = {
    'ag_abi_version': (C.c_int, []),
    'ag_bce_logits_fwd_strided': (C.c_int, [vp, i64, i64, C.c_float, vp, vp, vp, vp, C.c_float, C.c_int, C.c_int, vp]),
    'ag_bce_logits_bwd_strided': (C.c_int, [vp, i64, i64, C.c_float, vp, vp, vp, C.c_float, vp, i64, i64, C.c_int, C.c_int, vp]),
    'ag_transpose_batched': (C.c_int, [vp, C.c_int, i64, i64, vp, C.c_int, i64, i64, C.c_int, C.c_int, C.c_int, vp]),
    'ag_defer_reduces': (C.c_int, [C.c_int]),
    'ag_flush_reduces': (C.c_int, [vp]),
    'ag_arch': (C.c_char_p, []),
    'ag_last_error': (C.c_char_p, []),
    'ag_weight_norm_fwd': (C.c_int, [vp, C.c_int, C.c_int, vp]),
    'ag_weight_norm_bwd': (C.c_int, [vp, C.c_int, C.c_int, vp]),
    'ag_conv1d_engine': (C.c_int, [C.POINTER(ConvArgs), vp]),
    'ag_prep_conv_weight': (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'ag_wpa_numel': (i64, [C.c_int, C.c_int, C.c_int]),
    'ag_wpb_numel': (i64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'ag_conv1d_wgrad': (C.c_int, [vp, i64, i64, vp, i64, i64, vp] + [C.c_int] * 8 + [vp]),
    'ag_conv1d_o1_fwd': (C.c_int, [vp, i64, i64, vp, vp, vp, i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, f32, vp]),
    'ag_conv1d_o1_bwd_data': (C.c_int, [vp, i64, vp, vp, i64, i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, vp]),
    'ag_conv1d_o1_wgrad': (C.c_int, [vp, i64, vp, i64, i64, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'ag_channel_sum': (C.c_int, [vp, i64, i64, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'ag_lstm_front_bwd_step': (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, vp,
                                         vp, vp, vp, C.c_int, C.c_int, vp]),
    'ag_leaky_bwd': (C.c_int, [vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, vp, C.c_int,
                               C.c_int, C.c_int, f32, vp]),
    'ag_gemm': (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int,
                          C.c_int, C.c_int, f32, f32, vp, vp, C.c_int, C.c_int, f32, vp]),
    'ag_col_sum': (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp]),
    'ag_lstm_cell_fwd': (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int,
                                   vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp]),
    'ag_lstm_cell_bwd': (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int,
                                   vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int,
                                   C.c_int, C.c_int, vp]),
    'ag_gru_cell_fwd': (C.c_int, [vp, vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp]),
    'ag_gru_cell_bwd': (C.c_int, [vp, vp, vp, C.c_int, vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    'ag_skinny_gemm': (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                 f32, vp, C.c_int, f32, C.c_int, vp]),
    'ag_lstm_step_fwd': (C.c_int, [vp, vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_int,
                                   C.c_int, C.c_int, vp]),
    'ag_lstm_seq_fwd': (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'ag_lstm_seq_bwd': (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'ag_set_precision': (C.c_int, [C.c_int]),
    'ag_get_precision': (C.c_int, []),
    'ag_bind_workspace': (C.c_int, [vp, i64]),
    'ag_conv1d_wgrad_ws_numel': (i64, [C.c_int] * 5),
    'ag_gemm_ws_numel': (i64, [C.c_int] * 4),
    'ag_skinny_ws_numel': (i64, [C.c_int] * 3),
    'ag_persist_debug': (C.c_int, [i64, C.c_int]),
    'ag_lstm_persist_ok': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'ag_lstm_persist_ws_bytes': (i64, [C.c_int, C.c_int, C.c_int]),
    'ag_lstm_seq_fwd_persist': (C.c_int, [vp, vp, vp, vp, C.c_int, vp, vp, vp, i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'ag_lstm_persist_bwd_ok': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'ag_lstm_seq_bwd_persist': (C.c_int, [vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'ag_rowdot_fwd': (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp, i64, C.c_int, C.c_int, vp]),
    'ag_rowdot_bwd_ws_numel': (i64, [C.c_int, C.c_int]),
    'ag_rowdot_bwd': (C.c_int, [vp, i64, vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, f32, vp]),
    'ag_gemm_h_ok': (C.c_int, [C.c_int] * 7),
    'ag_gemm_h_ws_numel': (i64, [C.c_int] * 5),
    'ag_gemm_h': (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, f32, f32, vp,
                            vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, f32, vp]),
    'ag_to_bf16_2d': (C.c_int, [vp, i64, vp, i64, C.c_int, C.c_int, vp]),
    'ag_gfront_persist_ok': (C.c_int, [C.c_int] * 4),
    'ag_gfront_persist_ws_bytes': (i64, [C.c_int] * 3),
    'ag_gfront_bwd_persist_ok': (C.c_int, [C.c_int] * 4),
    'ag_gfront_bwd_persist': (C.c_int, [vp, vp, vp, i64, vp, vp, i64, vp, vp, C.c_int, vp, vp, vp, vp, i64] + [C.c_int] * 5 + [vp]),
    'ag_grufront_bwd_persist': (C.c_int, [vp, vp, vp, vp, i64, vp, vp, i64, vp, vp, C.c_int, vp, vp, vp, vp, vp, i64] + [C.c_int] * 5 + [vp]),
    'ag_gfront_fwd_persist': (C.c_int, [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, i64, vp, vp, i64] + [C.c_int] * 5 + [vp]),
    'ag_grufront_fwd_persist': (C.c_int, [vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, i64, vp, vp, i64] + [C.c_int] * 5 + [vp]),
    'ag_build_zc': (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'ag_critic_batch': (C.c_int, [vp, i64, vp, i64, C.c_int, vp, i64, vp, i64, C.c_int, C.c_int, vp, vp, vp,
                                  C.POINTER(i32), C.c_int, vp, vp, vp, C.c_int, vp, vp]),
    'ag_convlstm_peephole_fwd': (C.c_int, [vp] * 8 + [C.c_int] * 4 + [vp]),
    'ag_convlstm_peephole_bwd': (C.c_int, [vp] * 11 + [C.c_int] * 4 + [vp]),
    'ag_convlstm_cell_fwd': (C.c_int, [vp] * 6 + [f32, vp, vp] + [C.c_int] * 4 + [vp]),
    'ag_convlstm_cell_bwd': (C.c_int, [vp] * 6 + [f32] + [vp] * 7 + [C.c_int] * 4 + [vp]),
    'ag_convlstm_out_fwd': (C.c_int, [vp, vp, vp, i64, vp]),
    'ag_convlstm_out_bwd': (C.c_int, [vp, vp, vp, vp, vp, i64, vp]),
    'ag_layer_norm_hbfw_fwd': (C.c_int, [vp, vp, vp, f32, vp, vp, vp] + [C.c_int] * 4 + [vp]),
    'ag_layer_norm_hbfw_bwd': (C.c_int, [vp] * 8 + [C.c_int] * 4 + [vp]),
    'ag_bce_logits_fwd': (C.c_int, [vp, C.c_int, f32, vp, vp, vp, f32, C.c_int, C.c_int, vp]),
    'ag_bce_logits_bwd': (C.c_int, [vp, C.c_int, f32, vp, vp, f32, vp, C.c_int, C.c_int, C.c_int, vp]),
    'ag_act_fwd': (C.c_int, [vp, vp, i64, C.c_int, f32, vp]),
    'ag_act_bwd': (C.c_int, [vp, vp, vp, i64, C.c_int, f32, vp]),
    'ag_time_moments_fwd': (C.c_int, [vp, i64, i64, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    'ag_time_moments_bwd': (C.c_int, [vp, i64, i64, vp, vp, vp, vp, vp, i64, i64, C.c_int, C.c_int, C.c_int, vp]),
    'ag_act_bwd2d': (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, f32, vp]),
    'ag_axpby': (C.c_int, [vp, vp, i64, f32, f32, vp]),
    'ag_grad_norms': (C.c_int, [vp, C.c_int, vp, vp, vp, f32, vp, C.c_int, vp]),
    'ag_opt_step': (C.c_int, [vp, C.c_int, vp, C.c_int, f32, f32, f32, f32, f32, f32, C.c_int, vp, vp, vp, vp, vp, vp]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'audiogan_amd: %s not found. Build it with `python -c "import __graft_entry__ as g; '
            'g.build()"` or `make -C audiogan_amd/csrc`. There is no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    return lib


ABI_VERSION = 10      # what this package was written against (csrc/api.hip: ag_abi_version)

lib = _load()
if lib.ag_abi_version() != ABI_VERSION:
    raise ImportError('audiogan_amd: %s has ABI version %d, this package needs %d - rebuild it (make -C audiogan_amd/csrc)'
                      % (LIB_PATH, lib.ag_abi_version(), ABI_VERSION))


class AudioganHipError(RuntimeError):
    pass


def check(rc, what):
    """Map C status codes to Python exceptions (SURVEY.md 8(b) error convention)."""
    if rc == AG_OK:
        return
    msg = lib.ag_last_error().decode() or what
    if rc == AG_ERR_ARG:
        raise ValueError(msg)
    if rc == AG_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise AudioganHipError(msg)
