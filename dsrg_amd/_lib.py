"""ctypes binding of libdsrg_hip.so (include/dsrg_hip.h).  There is no fallback: if the library
is missing or a call fails, an exception is raised."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DSRG_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libdsrg_hip.so')

F32, BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_RELU, ACT_TANH = 0, 1, 2, 3
MASK_NONE, MASK_NEG10000, MASK_NEGINF, MASK_PLUS1 = 0, 1, 2, 3
IDX_I32, IDX_I64, IDX_F32 = 0, 1, 2
MAX_GROUPS = 8
ABI_VERSION = 14


class HeadTailProblem(C.Structure):
    _fields_ = [('h', C.c_void_p), ('gamma', C.c_void_p), ('beta', C.c_void_p), ('w2', C.c_void_p), ('b2', C.c_void_p),
                ('score', C.c_void_p), ('M', C.c_int32), ('ldh', C.c_int32)]


class TransposeProblem(C.Structure):
    _fields_ = [('inp', C.c_void_p), ('out', C.c_void_p), ('colsum', C.c_void_p), ('ld_in', C.c_int32), ('ld_out', C.c_int32),
                ('R', C.c_int32), ('C', C.c_int32), ('Rp', C.c_int32), ('in_dtype', C.c_int32), ('out_dtype', C.c_int32), ('pad_', C.c_int32)]


MAX_TRANSPOSES = 24


class GemmProblem(C.Structure):
    _fields_ = [('A', C.c_void_p), ('W', C.c_void_p), ('bias', C.c_void_p), ('residual', C.c_void_p),
                ('C', C.c_void_p), ('M', C.c_int32), ('N', C.c_int32), ('K', C.c_int32),
                ('lda', C.c_int32), ('ldw', C.c_int32), ('ldr', C.c_int32), ('ldc', C.c_int32), ('reserved0_', C.c_int32),
                ('a_stats', C.c_void_p), ('a_c1', C.c_void_p), ('r_stats', C.c_void_p), ('r_gamma', C.c_void_p),
                ('r_beta', C.c_void_p), ('stats_out', C.c_void_p), ('a_np', C.c_int32), ('r_np', C.c_int32),
                ('ln_eps', C.c_float), ('trans', C.c_int32), ('a_colsum', C.c_void_p), ('c_pre', C.c_void_p), ('ldcp', C.c_int32), ('res_act', C.c_int32)]


class AttnProblem(C.Structure):
    _fields_ = [('Q', C.c_void_p), ('K', C.c_void_p), ('V', C.c_void_p), ('O', C.c_void_p),
                ('q_bs', C.c_int64), ('k_bs', C.c_int64), ('v_bs', C.c_int64), ('o_bs', C.c_int64),
                ('q_rs', C.c_int32), ('k_rs', C.c_int32), ('v_rs', C.c_int32), ('o_rs', C.c_int32),
                ('kmask', C.c_void_p), ('pair_dists', C.c_void_p), ('lse', C.c_void_p), ('Nq', C.c_int32), ('Nk', C.c_int32),
                ('klens', C.c_void_p), ('kseg', C.c_int32), ('klens2', C.c_void_p)]


ATTN_MAX_GROUPS = 4


class EmbedBwdArgs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('base', 'feat', 'Wf', 'bf', 'lnf_gamma', 'lnf_beta', 'E1', 'idx1', 'E2', 'idx2', 'vec',
                                          'lno_gamma', 'lno_beta', 'dout', 'dbase', 'dWf', 'dbf', 'dlnf_gamma', 'dlnf_beta', 'dE1',
                                          'dE2', 'dvec', 'dlno_gamma', 'dlno_beta')] + \
               [(n, C.c_int32) for n in ('ldb', 'ldf', 'F', 'idx1_kind', 'idx1_stride', 'idx1_div', 'idx2_kind', 'idx2_stride',
                                         'idx2_div', 'lddo', 'lddb', 'M', 'H', 'dtype')] + [('eps', C.c_float), ('E1_rows', C.c_int32),
                                                                                   ('E2_rows', C.c_int32), ('E1_padding_idx', C.c_int32), ('E1_copies', C.c_int32),
                                                                                   ('reserved_', C.c_int32), ('partials', C.c_void_p)]


class GraphState(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('pos', 'dist', 'point', 'visited', 'step_id', 'emb_sum', 'emb_cnt', 'n_nodes')] + \
               [('N', C.c_int32), ('H', C.c_int32)]


class GraphNavOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('gmap_img', 'gmap_step_ids', 'gmap_pos_fts', 'gmap_pair_dists', 'gmap_masks', 'gmap_visited',
                                          'gmap_slots', 'gmap_src', 'bw_mask', 'vp_pos_fts', 'no_vp_left', 'gmap_len')] + \
               [('G', C.c_int32), ('Vp', C.c_int32)]


class LnProblem(C.Structure):
    _fields_ = [('x', C.c_void_p), ('residual', C.c_void_p), ('gamma', C.c_void_p), ('beta', C.c_void_p), ('out', C.c_void_p),
                ('M', C.c_int32), ('ldx', C.c_int32), ('ldr', C.c_int32), ('ldo', C.c_int32),
                ('o_seg', C.c_int32), ('o_seg_ld', C.c_int32), ('aux0', C.c_void_p), ('aux1', C.c_void_p),
                ('aux_seg', C.c_int32), ('aux_ld0', C.c_int32), ('aux_ld1', C.c_int32), ('reserved_', C.c_int32)]


_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
SIGNATURES = {
    'dsrg_abi_version': (C.c_int, []),
    'dsrg_init': (C.c_int, []),
    'dsrg_set_dropout_offset': (C.c_int, [_P]),
    'dsrg_kl_rows_fwd': (C.c_int, [_P, _I, _P, _I, _P, _P, _I, _I, _P]),
    'dsrg_kl_rows_bwd': (C.c_int, [_P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _P]),
    'dsrg_sum_slabs': (C.c_int, [_P, _I, _L, _P, _L, _I, _P]),
    'dsrg_dropout_add': (C.c_int, [_P, _P, _P, _L, _I, _F, C.c_uint64, _P]),
    'dsrg_opt_chunk': (C.c_int, []),
    'dsrg_grad_sumsq': (C.c_int, [_P, _P, _I, _P, _P]),
    'dsrg_adamw_prepare': (C.c_int, [_P, _I, _F, _F, _P, _P, _I, _P, _P]),
    'dsrg_adamw_step': (C.c_int, [_P, _P, _I, _P, _P, _P, C.c_double, C.c_double, C.c_double, _I, _P]),
    'dsrg_last_error': (C.c_char_p, []),
    'dsrg_gemm_nt': (C.c_int, [C.POINTER(GemmProblem), _I, _I, _I, _I, _P]),
    'dsrg_gemm_nt_tiled': (C.c_int, [C.POINTER(GemmProblem), _I, _I, _I, _I, _I, _I, _I, _P]),
    'dsrg_gemm_nt_plan': (C.c_int, [C.POINTER(GemmProblem), _I, _I, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'dsrg_attention_fwd': (C.c_int, [_P, _L, _I, _P, _L, _I, _P, _L, _I, _P, _L, _I, _P, _I, _P, _P, _P,
                                     _I, _I, _I, _I, _F, _P, _F, C.c_uint64, _I, _P]),
    'dsrg_attention_fwd_grouped': (C.c_int, [C.POINTER(AttnProblem), _I, _I, _P, _P, _I, _I, _F, _F, C.c_uint64, _I, _P]),
    'dsrg_attention_bwd': (C.c_int, [_P, _L, _I, _P, _L, _I, _P, _L, _I, _P, _L, _I, _P, _L, _I, _P, _P, _I, _P, _P, _P,
                                     _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, C.c_uint64, _I, _P]),
    'dsrg_attention_bwd_strided': (C.c_int, [_P, _L, _I, _P, _L, _I, _P, _L, _I, _P, _L, _I, _P, _L, _I, _P, _P, _I, _P, _P, _P,
                                             _P, _L, _I, _P, _L, _I, _P, _L, _I, _P, _P, _I, _I, _I, _I, _F, _F, C.c_uint64, _I, _P]),
    'dsrg_attention_bwd_noalloc': (C.c_int, [_P, _L, _I, _P, _L, _I, _P, _L, _I, _P, _L, _I, _P, _L, _I, _P, _P, _I, _P, _P, _P,
                                             _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, C.c_uint64, _I, _P]),
    'dsrg_layernorm_fwd': (C.c_int, [_P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _F, _I, _I, _P]),
    'dsrg_layernorm_fwd_grouped': (C.c_int, [C.POINTER(LnProblem), _I, _I, _F, _I, _I, _P]),
    'dsrg_layernorm_fwd_grouped_drop': (C.c_int, [C.POINTER(LnProblem), _I, _I, _F, _I, _I, _F, C.c_uint64, _P]),
    'dsrg_layernorm_bwd_drop': (C.c_int, [_P, _I, _P, _I, _P, _P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _F, _I, _I, _F, C.c_uint64, _I, _P]),
    'dsrg_layernorm_bwd_blocks': (C.c_int, [_I]),
    'dsrg_embed_ln_fwd': (C.c_int, [_P, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _I, _I, _I,
                                    _P, _P, _P, _F, _P, _I, _I, _I, _I, _P]),
    'dsrg_embed_ln_fwd_seg': (C.c_int, [_P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _I, _I, _I,
                                        _P, _P, _P, _F, _P, _I, _I, _I, _I, _P]),
    'dsrg_embed_ln_fwd_lnb': (C.c_int, [_P, _I, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _I, _I, _I,
                                        _P, _P, _P, _F, _P, _I, _I, _I, _I, _P]),
    'dsrg_group_mean_fwd': (C.c_int, [_P, _I, _P, _I, _I, _I, _I, _I, _P]),
    'dsrg_gaa_pool_fwd': (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'dsrg_gaa_pool_fwd_ex': (C.c_int, [_P, _L, _P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _P]),
    'dsrg_gaa_pool_fwd_ln': (C.c_int, [_P, _L, _P, _P, _F, _P, _L, _P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _P]),
    'dsrg_head_tail_fwd': (C.c_int, [_P, _I, _P, _P, _P, _P, _F, _P, _I, _I, _P]),
    'dsrg_nav_logits_fwd': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'dsrg_nav_logits_fwd_og': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'dsrg_transpose_pad': (C.c_int, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    'dsrg_head_tail_fwd_grouped': (C.c_int, [C.POINTER(HeadTailProblem), _I, _I, C.c_float, _P]),
    'dsrg_transpose_pad_grouped': (C.c_int, [C.POINTER(TransposeProblem), _I, _P]),
    'dsrg_colsum_acc': (C.c_int, [_P, _I, _P, _I, _I, _I, _P]),
    'dsrg_act_fwd': (C.c_int, [_P, _P, _L, _I, _I, _P]),
    'dsrg_act_bwd': (C.c_int, [_P, _P, _P, _L, _I, _I, _P]),
    'dsrg_layernorm_bwd': (C.c_int, [_P, _I, _P, _I, _P, _P, _I, _P, _I, _P, _P, _P, _I, _I, _F, _I, _I, _P]),
    'dsrg_group_mean_bwd': (C.c_int, [_P, _I, _P, _I, _I, _I, _I, _I, _P]),
    'dsrg_gaa_pool_bwd': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'dsrg_head_tail_bwd': (C.c_int, [_P, _I, _P, _P, _P, _P, _F, _P, _I, _P, _P, _P, _P, _I, _I, _P]),
    'dsrg_nav_logits_bwd': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'dsrg_igl_blend_bwd': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'dsrg_embed_ln_bwd': (C.c_int, [C.POINTER(EmbedBwdArgs), _P]),
    'dsrg_embed_ln_bwd_workspace': (C.c_int64, [_P]),
    'dsrg_segment_wsum': (C.c_int, [_P, _I, _I, _P, _I, _I, _P, _I, _I, _P, _I, _I, _P, _P, _P, _I, _I, _P]),
    'dsrg_graph_reset': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'dsrg_graph_observe': (C.c_int, [C.POINTER(GraphState), _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'dsrg_graph_update_embeds': (C.c_int, [C.POINTER(GraphState), _P, _P, _P, _P, _I, _P, _L, _I, _P, _I, _I, _I, _P]),
    'dsrg_graph_nav_inputs': (C.c_int, [C.POINTER(GraphState), C.POINTER(GraphNavOut), _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _P]),
    'dsrg_action_tail_fwd': (C.c_int, [_P, _I, _P, _L, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    'dsrg_ce_rows_fwd': (C.c_int, [_P, _I, _P, _L, _P, _P, _I, _I, _P]),
    'dsrg_ce_rows_bwd': (C.c_int, [_P, _I, _P, _L, _P, _P, _P, _I, _I, _I, _I, _P]),
    'dsrg_igl_blend_fwd': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
}

_lib = None


class DsrgHipError(RuntimeError):
    pass


def load():
    """Load the shared library and bind every symbol of include/dsrg_hip.h.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DsrgHipError('%s not found: build it with `python -m dsrg_amd.build` (hipcc, gfx950); '
                           'there is no non-HIP fallback for the DSRG hot path' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype, fn.argtypes = res, args
    v = lib.dsrg_abi_version()
    if v != ABI_VERSION:
        raise DsrgHipError('libdsrg_hip.so ABI version %d, binding expects %d' % (v, ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise DsrgHipError('%s failed (%d): %s' % (what, rc, load().dsrg_last_error().decode()))
