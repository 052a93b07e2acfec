"""ctypes bindings of libc1dvae.so (declared in include/c1dvae.h).

The product path FAILS LOUDLY when the shared library is missing: there is no CPU or eager-PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('C1V_LIB') or os.path.join(_HERE, 'libc1dvae.so')      # C1V_LIB: diagnostic builds only

F32 = 0
BF16 = 1
SLACK_ROWS = 72
CHUNK = 2048
PACK_TILE = 64          # C1V_PACK_TILE
ROW_STORE, ROW_POOL2, ROW_UP2, ROW_PAIRSUM, ROW_UNPOOL, ROW_LATENT_BWD, ROW_NORM = 0, 1, 2, 3, 4, 5, 6


def fused_width(kp: int) -> bool:
    """Row widths (elements) the fused bf16 stream kernel is instantiated for (vs_fused_width in c1v_loss.hip)."""
    return kp == 32 or (kp % 64 == 0 and 64 <= kp <= 320)


class Seq(C.Structure):
    _fields_ = [('data', C.c_void_p), ('dtype', C.c_int32), ('batch', C.c_int32), ('bstride', C.c_int64),
                ('off', C.c_int32), ('len', C.c_int32), ('cp', C.c_int32), ('c', C.c_int32)]


class Epilogue(C.Structure):
    _fields_ = [('bias', C.c_void_p), ('relu', C.c_int32), ('row_mode', C.c_int32), ('reshape_w', C.c_int32),
                ('dst', Seq), ('act', Seq), ('act_rowmul', C.c_int32), ('act_scale', C.c_float),
                ('sel', C.c_void_p), ('sel_bstride', C.c_int64), ('sel_cp', C.c_int32),
                ('keep', C.c_void_p), ('keep_scale', C.c_float),
                ('lat_mlv', C.c_void_p), ('lat_eps', C.c_void_p), ('lat_keep', C.c_void_p), ('lat_D', C.c_int32),
                ('lat_keep_prob', C.c_float), ('lat_std_scale', C.c_float), ('lat_kc', C.c_float), ('bias_row_w', C.c_int32),
                ('nrm_yhat', C.c_void_p), ('nrm_kp', C.c_int32), ('nrm_rnorm', C.c_void_p), ('nrm_qscale', C.c_float),
                ('nrm_pos', C.c_void_p), ('nrm_live_only', C.c_int32), ('nrm_dy', Seq)]


class ConvArgs(C.Structure):
    _fields_ = [('x', Seq), ('a_off', C.c_int32), ('n_rows', C.c_int32), ('k', C.c_int32), ('w_op', C.c_void_p),
                ('ldw', C.c_int64), ('n_out', C.c_int32), ('epi', Epilogue)]


class DenseArgs(C.Structure):
    _fields_ = [('a', C.c_void_p), ('dtype', C.c_int32), ('rows', C.c_int32), ('lda', C.c_int64), ('K', C.c_int32),
                ('w_op', C.c_void_p), ('ldw', C.c_int64), ('n_out', C.c_int32), ('epi', Epilogue), ('k_splits', C.c_int32),
                ('split_stride', C.c_int64)]


class WgradArgs(C.Structure):
    _fields_ = [('a', C.c_void_p), ('lda', C.c_int64), ('I', C.c_int32),
                ('d', C.c_void_p), ('ldd', C.c_int64), ('J', C.c_int32),
                ('dtype', C.c_int32), ('r_begin', C.c_int64), ('r_end', C.c_int64), ('a_shift0', C.c_int32),
                ('taps', C.c_int32), ('flip', C.c_int32), ('tap_stride', C.c_int64), ('si', C.c_int64),
                ('sj', C.c_int64), ('out', C.c_void_p), ('slab_stride', C.c_int64), ('splits', C.c_int32),
                ('bias_out', C.c_void_p), ('bias_slab_stride', C.c_int64), ('tap_group', C.c_int32)]


class BwdPairArgs(C.Structure):
    _fields_ = [('conv_dgrad', C.POINTER(ConvArgs)), ('dense_dgrad', C.POINTER(DenseArgs)), ('wgrad', C.POINTER(WgradArgs))]


class ColsumArgs(C.Structure):
    _fields_ = [('buf', C.c_void_p), ('dtype', C.c_int32), ('rows', C.c_int64), ('ld', C.c_int64), ('C', C.c_int32),
                ('out', C.c_void_p), ('slab_stride', C.c_int64), ('splits', C.c_int32)]


class PackDesc(C.Structure):
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('dtype', C.c_int32), ('n_count', C.c_int32),
                ('taps', C.c_int32), ('c_count', C.c_int32), ('cp', C.c_int32), ('flip', C.c_int32),
                ('ldk', C.c_int64), ('s_tap', C.c_int64), ('s_c', C.c_int64), ('s_n', C.c_int64),
                ('first_block', C.c_int32), ('pad', C.c_int32)]


class PackInputArgs(C.Structure):
    _fields_ = [('src', C.c_void_p), ('T', C.c_int32), ('C', C.c_int32), ('dst', Seq)]


class GatherInputArgs(C.Structure):
    _fields_ = [('table', C.c_void_p), ('ld', C.c_int64), ('ids', C.c_void_p), ('T', C.c_int32), ('C', C.c_int32),
                ('dst', Seq)]


class LatentArgs(C.Structure):
    _fields_ = [('mlv', C.c_void_p), ('B', C.c_int32), ('D', C.c_int32), ('eps', C.c_void_p), ('keep_d', C.c_void_p),
                ('keep_prob', C.c_float), ('std_scale', C.c_float), ('zd_out', C.c_void_p), ('dtype', C.c_int32),
                ('ldz', C.c_int64), ('z_out', C.c_void_p), ('kl_out', C.c_void_p), ('dzd', C.c_void_p),
                ('B_global', C.c_int32), ('dmlv', C.c_void_p), ('ldd', C.c_int64), ('n_parts', C.c_int32),
                ('part_stride', C.c_int64)]


BN_POOL2, BN_UP2 = 1, 2
BN_BLOCKS = 128          # partial-sum blocks of c1v_bn_reduce


class BnArgs(C.Structure):
    _fields_ = [('x', Seq), ('mode', C.c_int32), ('grad', Seq), ('sel', C.c_void_p), ('sel_bstride', C.c_int64),
                ('sel_cp', C.c_int32), ('keep', C.c_void_p), ('keep_scale', C.c_float), ('mean', C.c_void_p),
                ('var', C.c_void_p), ('eps', C.c_float), ('gamma', C.c_void_p), ('beta', C.c_void_p), ('sums', C.c_void_p),
                ('n_global', C.c_int64), ('dst', Seq), ('sel_out', C.c_void_p)]


class L2NormArgs(C.Structure):
    _fields_ = [('y', C.c_void_p), ('M', C.c_int32), ('V', C.c_int32), ('ldy', C.c_int64), ('yhat', C.c_void_p),
                ('dtype', C.c_int32), ('kp', C.c_int32), ('rnorm', C.c_void_p), ('qscale', C.c_float), ('pos', C.c_void_p)]


class XentAccumArgs(C.Structure):
    _fields_ = [('yhat', C.c_void_p), ('dtype', C.c_int32), ('M', C.c_int32), ('kp', C.c_int32), ('w', C.c_void_p),
                ('N', C.c_int32), ('n_pad', C.c_int32), ('ut', C.c_void_p), ('m_pad', C.c_int32),
                ('splits', C.c_int32), ('need_u', C.c_int32), ('ones_col', C.c_int32), ('logits', C.c_void_p),
                ('n_chunk', C.c_int32), ('live_ctl', C.c_void_p), ('ut_dtype', C.c_int32)]


class XentFinishArgs(C.Structure):
    _fields_ = [('ut', C.c_void_p), ('splits', C.c_int32), ('kp', C.c_int32), ('m_pad', C.c_int32),
                ('ones_col', C.c_int32), ('yhat', C.c_void_p), ('w', C.c_void_p), ('dtype', C.c_int32),
                ('rnorm', C.c_void_p), ('qscale', C.c_float), ('ids', C.c_void_p), ('B', C.c_int32), ('T', C.c_int32),
                ('V', C.c_int32), ('B_global', C.c_int32), ('xent_rows', C.c_void_p), ('xent_b', C.c_void_p),
                ('dy', Seq), ('y', C.c_void_p), ('ldy', C.c_int64), ('pos', C.c_void_p), ('live_ctl', C.c_void_p),
                ('ut_dtype', C.c_int32), ('live_rows', C.c_void_p), ('live_ids', C.c_void_p)]


class SegDesc(C.Structure):
    _fields_ = [('slab', C.c_void_p), ('slab_stride', C.c_int64), ('splits', C.c_int32), ('reps', C.c_int32),
                ('rep_stride', C.c_int64), ('offset', C.c_int64), ('count', C.c_int64), ('frozen', C.c_int32),
                ('first_block', C.c_int32)]


class LossReduceArgs(C.Structure):
    _fields_ = [('xent_rows', C.c_void_p), ('ids', C.c_void_p), ('kl_b', C.c_void_p), ('B', C.c_int32), ('T', C.c_int32),
                ('B_global', C.c_int32), ('out3', C.c_void_p), ('xent_b', C.c_void_p), ('pre_summed', C.c_int32)]


WGRAD_BATCH_MAX = 12
LIST_SIDE, LIST_FORK, LIST_JOIN = 1, 2, 4


class WgradBatchArgs(C.Structure):
    _fields_ = [('n', C.c_int32), ('items', C.POINTER(WgradArgs) * WGRAD_BATCH_MAX)]


class DecHeadArgs(C.Structure):
    _fields_ = [('latent', C.POINTER(LatentArgs)), ('dense', C.POINTER(DenseArgs)), ('deconv', C.POINTER(ConvArgs))]


class GradFinalizeArgs(C.Structure):
    _fields_ = [('segs', C.c_void_p), ('n_seg', C.c_int32), ('total_blocks', C.c_int32), ('grad', C.c_void_p),
                ('sq_partial', C.c_void_p), ('loss', C.c_void_p)]


class NadamArgs(C.Structure):
    _fields_ = [('p', C.c_void_p), ('g', C.c_void_p), ('m', C.c_void_p), ('v', C.c_void_p), ('n', C.c_int64),
                ('grad_scale', C.c_float),
                ('clipnorm', C.c_float), ('sched', C.c_void_p), ('beta1', C.c_float), ('beta2', C.c_float),
                ('eps', C.c_float), ('norm_out', C.c_void_p), ('sq_partial', C.c_void_p), ('n_partial', C.c_int32),
                ('sched_out7', C.c_void_p), ('ctr', C.c_void_p)]


class RngDesc(C.Structure):
    _fields_ = [('out', C.c_void_p), ('n', C.c_int64), ('kind', C.c_int32), ('thr', C.c_uint32),
                ('stream_id', C.c_uint64), ('offset', C.c_uint64), ('ctr_mul', C.c_uint64),
                ('first_block', C.c_int32), ('pad', C.c_int32)]


class LiveRowsArgs(C.Structure):
    _fields_ = [('ids', C.c_void_p), ('M', C.c_int32), ('items', C.c_int32), ('vocab_tiles', C.c_int32),
                ('max_splits', C.c_int32), ('pos', C.c_void_p), ('ctl', C.c_void_p), ('rows', C.c_void_p), ('cids', C.c_void_p)]


class StepPrologueArgs(C.Structure):
    _fields_ = [('rng_descs', C.c_void_p), ('n_rng', C.c_int32), ('rng_blocks', C.c_int32), ('seed', C.c_uint64),
                ('ctr', C.c_void_p), ('pack', C.POINTER(PackInputArgs)), ('gather', C.POINTER(GatherInputArgs)),
                ('live', C.POINTER(LiveRowsArgs)), ('nadam_state', C.c_void_p), ('sched', C.c_void_p),
                ('pack_descs', C.c_void_p), ('n_pack', C.c_int32), ('pack_blocks', C.c_int32)]


class VocabArgmaxArgs(C.Structure):
    _fields_ = [('q', C.c_void_p), ('dtype', C.c_int32), ('M', C.c_int32), ('kp', C.c_int32), ('w', C.c_void_p),
                ('N', C.c_int32), ('n_pad', C.c_int32), ('n_base', C.c_int32), ('part_score', C.c_void_p),
                ('part_idx', C.c_void_p), ('m_pad', C.c_int32), ('splits', C.c_int32), ('score', C.c_void_p),
                ('idx', C.c_void_p), ('logits', C.c_void_p), ('n_chunk', C.c_int32)]


class VocabArgmaxNoallocArgs(C.Structure):
    _fields_ = [('q', C.c_void_p), ('M', C.c_int32), ('kp', C.c_int32), ('w', C.c_void_p), ('N', C.c_int32),
                ('n_pad', C.c_int32), ('n_base', C.c_int32), ('score', C.c_void_p), ('idx', C.c_void_p)]


NOALLOC_SCRATCH_BYTES = 8 << 20
NOALLOC_MAX_QUERIES = NOALLOC_SCRATCH_BYTES // 8 // 128 * 128


class VocabTopnArgs(C.Structure):
    _fields_ = [('q', C.c_void_p), ('dtype', C.c_int32), ('M', C.c_int32), ('kp', C.c_int32), ('w', C.c_void_p),
                ('N', C.c_int32), ('n_base', C.c_int32), ('logits', C.c_void_p), ('ld', C.c_int64),
                ('topn', C.c_int32), ('score', C.c_void_p), ('idx', C.c_void_p)]


# every symbol include/c1dvae.h declares: name -> (argtypes, restype)
_P = C.POINTER
SYMBOLS = {
    'c1v_conv1d_fwd': ([_P(ConvArgs), C.c_void_p], C.c_int),
    'c1v_conv1d_dgrad': ([_P(ConvArgs), C.c_void_p], C.c_int),
    'c1v_deconv1d_fwd': ([_P(ConvArgs), C.c_void_p], C.c_int),
    'c1v_deconv1d_dgrad': ([_P(ConvArgs), C.c_void_p], C.c_int),
    'c1v_dense_fwd': ([_P(DenseArgs), C.c_void_p], C.c_int),
    'c1v_dense_dgrad': ([_P(DenseArgs), C.c_void_p], C.c_int),
    'c1v_wgrad': ([_P(WgradArgs), C.c_void_p], C.c_int),
    'c1v_conv1d_wgrad': ([_P(WgradArgs), C.c_void_p], C.c_int),
    'c1v_deconv1d_wgrad': ([_P(WgradArgs), C.c_void_p], C.c_int),
    'c1v_dense_wgrad': ([_P(WgradArgs), C.c_void_p], C.c_int),
    'c1v_bwd_pair': ([_P(BwdPairArgs), C.c_void_p], C.c_int),
    'c1v_colsum': ([_P(ColsumArgs), C.c_void_p], C.c_int),
    'c1v_pack_weights': ([C.c_void_p, C.c_int32, C.c_int32, C.c_void_p], C.c_int),
    'c1v_pack_input': ([_P(PackInputArgs), C.c_void_p], C.c_int),
    'c1v_gather_input': ([_P(GatherInputArgs), C.c_void_p], C.c_int),
    'c1v_bn_reduce': ([_P(BnArgs), C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_float, C.c_void_p], C.c_int),
    'c1v_bn_stats': ([C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                      C.c_void_p], C.c_int),
    'c1v_bn_apply_fwd': ([_P(BnArgs), C.c_void_p], C.c_int),
    'c1v_bn_apply_bwd': ([_P(BnArgs), C.c_void_p], C.c_int),
    'c1v_latent_head_fwd': ([_P(LatentArgs), C.c_void_p], C.c_int),
    'c1v_latent_head_bwd': ([_P(LatentArgs), C.c_void_p], C.c_int),
    'c1v_l2norm_rows': ([_P(L2NormArgs), C.c_void_p], C.c_int),
    'c1v_live_rows': ([C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p], C.c_int),
    'c1v_live_rows_table': ([_P(LiveRowsArgs), C.c_void_p], C.c_int),
    'c1v_live_rows_gather': ([_P(LiveRowsArgs), _P(GatherInputArgs), C.c_void_p], C.c_int),
    'c1v_cos_softmax_accum': ([_P(XentAccumArgs), C.c_void_p], C.c_int),
    'c1v_cos_softmax_xent_fwd': ([_P(XentAccumArgs), C.c_void_p], C.c_int),
    'c1v_cos_softmax_xent_bwd': ([_P(XentAccumArgs), C.c_void_p], C.c_int),
    'c1v_cos_softmax_finish': ([_P(XentFinishArgs), C.c_void_p], C.c_int),
    'c1v_loss_reduce': ([C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                         C.c_void_p, C.c_void_p], C.c_int),
    'c1v_grad_finalize': ([C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p], C.c_int),
    'c1v_wgrad_batch': ([_P(WgradBatchArgs), C.c_void_p], C.c_int),
    'c1v_dec_head_fwd': ([_P(DecHeadArgs), C.c_void_p], C.c_int),
    'c1v_dec_head_fwd_check': ([_P(DecHeadArgs)], C.c_int),
    'c1v_loss_sample_means': ([_P(LossReduceArgs), C.c_void_p], C.c_int),
    'c1v_grad_finalize_step': ([_P(GradFinalizeArgs), C.c_void_p], C.c_int),
    'c1v_grad_finalize_loss': ([C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, _P(LossReduceArgs), C.c_void_p],
                               C.c_int),
    'c1v_sqnorm_partial': ([C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p], C.c_int),
    'c1v_nadam_step': ([_P(NadamArgs), C.c_void_p], C.c_int),
    'c1v_nadam_advance': ([C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p], C.c_int),
    'c1v_eos_truncate_normalize_ld': ([C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p],
                                      C.c_int),
    'c1v_eos_truncate_normalize': ([C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p],
                                   C.c_int),
    'c1v_eos_truncate_normalize_f64': ([C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_void_p], C.c_int),
    'c1v_vocab_argmax': ([_P(VocabArgmaxArgs), C.c_void_p], C.c_int),
    'c1v_vocab_argmax_noalloc': ([_P(VocabArgmaxNoallocArgs), C.c_void_p], C.c_int),
    'c1v_vocab_topn': ([_P(VocabTopnArgs), C.c_void_p], C.c_int),
    'c1v_argmax_merge': ([C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p], C.c_int),
    'c1v_philox_normal': ([C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64,
                           C.c_void_p], C.c_int),
    'c1v_philox_bernoulli': ([C.c_void_p, C.c_int64, C.c_float, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p,
                              C.c_uint64, C.c_void_p], C.c_int),
    'c1v_counter_add': ([C.c_void_p, C.c_uint64, C.c_void_p], C.c_int),
    'c1v_philox_multi': ([C.c_void_p, C.c_int32, C.c_int32, C.c_uint64, C.c_void_p, C.c_void_p], C.c_int),
    'c1v_step_prologue': ([_P(StepPrologueArgs), C.c_void_p], C.c_int),
    'c1v_run_list': ([C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p], C.c_int),
    'c1v_run_list_fork': ([C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p], C.c_int),
    'c1v_last_error': ([], C.c_char_p),
    'c1v_version': ([], C.c_int),
    'c1v_device_info': ([C.c_int32, _P(C.c_int32)], C.c_int),
}

_lib = None


class C1VError(RuntimeError):
    pass


def load():
    """Load libc1dvae.so and bind every declared symbol.  Raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise C1VError('libc1dvae.so is not built (%s). Run `python -c "import __graft_entry__ as g; g.build()"` or '
                       '`make -C conv1d_text_vae_amd/csrc`. There is no CPU fallback.' % LIB_PATH)
    # The HIP runtime must be the one PyTorch uses (streams and device pointers are shared with torch): load torch's
    # bundled libamdhip64 into the global symbol scope first; libc1dvae.so itself does not link a HIP runtime.
    import torch
    tlib = os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so')
    if os.path.isfile(tlib):
        C.CDLL(tlib, mode=C.RTLD_GLOBAL)
    else:
        C.CDLL('libamdhip64.so', mode=C.RTLD_GLOBAL)
    lib = C.CDLL(LIB_PATH)
    for name, (argtypes, restype) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = restype
    _lib = lib
    return lib


def check(rc: int, what: str = ''):
    if rc != 0:
        msg = load().c1v_last_error()
        raise C1VError('%s failed (%d): %s' % (what or 'libc1dvae call', rc, msg.decode() if msg else '?'))


def call(name: str, *args):
    lib = load()
    check(getattr(lib, name)(*args), name)
