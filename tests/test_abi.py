"""CPU-side checks of the drop-in boundary: libc1dvae.so loads without a GPU and exports every symbol that
include/c1dvae.h declares; the ctypes mirrors have the sizes the C structs have; the host-side checks reject bad
shapes before anything is launched (no compute call is made here)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conv1d_text_vae_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'c1dvae.h')


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(c1v_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), 'libc1dvae.so does not export %s' % n
        assert n in _lib.SYMBOLS, 'no ctypes binding for %s' % n
    for n in _lib.SYMBOLS:
        assert n in names, '%s is bound but not declared in include/c1dvae.h' % n
    assert lib.c1v_version() >= 100


def test_ctypes_struct_sizes_match_the_header(tmp_path):
    """Compile a tiny C program against the header and compare sizeof() of every argument struct."""
    structs = {'c1v_seq': _lib.Seq, 'c1v_epilogue': _lib.Epilogue, 'c1v_conv1d_args': _lib.ConvArgs,
               'c1v_dense_args': _lib.DenseArgs, 'c1v_wgrad_args': _lib.WgradArgs, 'c1v_colsum_args': _lib.ColsumArgs,
               'c1v_pack_desc': _lib.PackDesc, 'c1v_pack_input_args': _lib.PackInputArgs,
               'c1v_gather_input_args': _lib.GatherInputArgs, 'c1v_latent_args': _lib.LatentArgs,
               'c1v_l2norm_args': _lib.L2NormArgs, 'c1v_xent_accum_args': _lib.XentAccumArgs,
               'c1v_xent_finish_args': _lib.XentFinishArgs, 'c1v_seg_desc': _lib.SegDesc,
               'c1v_nadam_args': _lib.NadamArgs, 'c1v_vocab_argmax_args': _lib.VocabArgmaxArgs,
               'c1v_rng_desc': _lib.RngDesc, 'c1v_bn_args': _lib.BnArgs,
               'c1v_vocab_argmax_noalloc_args': _lib.VocabArgmaxNoallocArgs,
               'c1v_grad_finalize_args': _lib.GradFinalizeArgs, 'c1v_dec_head_args': _lib.DecHeadArgs,
               'c1v_wgrad_batch_args': _lib.WgradBatchArgs, 'c1v_live_rows_args': _lib.LiveRowsArgs,
               'c1v_loss_reduce_args': _lib.LossReduceArgs}
    prog = '#include <stdio.h>\n#include "c1dvae.h"\nint main(void){\n'
    for name in structs:
        prog += '  printf("%s %%zu\\n", sizeof(%s));\n' % (name, name)
    prog += '  printf("SLACK %d\\nCHUNK %d\\n", C1V_SLACK_ROWS, C1V_CHUNK);\n  return 0;}\n'
    src = tmp_path / 'sz.c'
    src.write_text(prog)
    exe = tmp_path / 'sz'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)]).decode().splitlines())
    for name, cls in structs.items():
        assert int(out[name]) == C.sizeof(cls), name
    assert int(out['SLACK']) == _lib.SLACK_ROWS and int(out['CHUNK']) == _lib.CHUNK


def test_host_side_checks_reject_bad_arguments_without_launching():
    lib = _lib.load()
    a = _lib.ConvArgs()                       # all zero: null pointers
    assert lib.c1v_conv1d_fwd(C.byref(a), None) != 0
    assert b'null' in lib.c1v_last_error()
    d = _lib.DenseArgs()
    assert lib.c1v_dense_fwd(C.byref(d), None) != 0
    w = _lib.WgradArgs()
    assert lib.c1v_wgrad(C.byref(w), None) != 0
    x = _lib.XentAccumArgs()
    assert lib.c1v_cos_softmax_accum(C.byref(x), None) != 0
    with pytest.raises(_lib.C1VError):
        _lib.check(-1, 'probe')
    # a pitch that is not a multiple of 32 elements is refused by the conv entry point
    buf = (C.c_char * 4096)()
    a.x = _lib.Seq(C.addressof(buf), _lib.BF16, 1, 40, 0, 2, 20, 20)
    a.w_op = C.addressof(buf)
    a.k, a.n_rows, a.n_out = 1, 2, 4
    assert lib.c1v_conv1d_fwd(C.byref(a), None) != 0
    assert b'multiple of 32' in lib.c1v_last_error()


def test_engine_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from conv1d_text_vae_amd import engine, geometry
    g = geometry.Geometry(12, 12, 10, 10, (6,), 3, 5)
    with pytest.raises(_lib.C1VError):
        engine.VAEEngine(g, 4)


def test_fused_decoder_head_check_accepts_and_refuses_on_the_host():
    """c1v_dec_head_fwd_check launches nothing: it only says whether the three linked ops are within the fused kernel's limits
    (the engine falls back to the three launches otherwise).  The C2 shapes qualify; other operand types, a dense operand that is
    not the latent head's output, too many taps x channels, and filter counts that are not whole 16-byte rows do not."""
    lib = _lib.load()
    L = _lib
    B, D, Dp, n_t, width, width_p, k, F, To = 256, 256, 256, 15, 18, 20, 3, 128, 32
    mem = (C.c_char * (1 << 16))()
    base = (C.addressof(mem) + 255) & ~255               # (only addresses and alignment are looked at)
    zd, xd, xf, wd, wc, bias = base, base + 4096, base + 8192, base + 12288, base + 16384, base + 20480

    def case(dtype=L.BF16, a=zd, cp=32, taps=k, n_out=F):
        lat = L.LatentArgs(base, B, D, base, None, 0.7, 1.0, zd, dtype, Dp, None, None, None, B, None, 0, 8, B * 2 * D)
        pitch = n_t + 3 * (taps - 1) + ((n_t + 3 * (taps - 1)) & 1)
        xd_seq = L.Seq(xd, dtype, B, pitch * cp, taps - 1, n_t, cp, width)
        e_d = L.Epilogue(bias, 1, L.ROW_STORE, width_p, xd_seq, L.Seq(), 1, 1.0, None, 0, 0, None, 1.0, bias_row_w=width)
        dense = L.DenseArgs(a, dtype, B, Dp, Dp, wd, Dp, n_t * width_p, e_d)
        xf_seq = L.Seq(xf, dtype, B, (To + 4) * 128, 1, To, 128, n_out)
        e_c = L.Epilogue(bias, 1, L.ROW_UP2, 0, xf_seq, L.Seq(), 1, 1.0, None, 0, 0, None, 1.0)
        conv = L.ConvArgs(xd_seq, 0, n_t + taps - 1, taps, wc, 128, n_out, e_c)
        keep = (lat, dense, conv)
        return L.DecHeadArgs(C.pointer(lat), C.pointer(dense), C.pointer(conv)), keep

    args, keep = case()
    assert lib.c1v_dec_head_fwd_check(C.byref(args)) == 0, lib.c1v_last_error()
    for kw, text in ((dict(dtype=L.F32), b'bf16'), (dict(a=zd + 512), b'latent'), (dict(cp=64), b'limits'), (dict(n_out=100), b'limits')):
        args, keep = case(**kw)
        assert lib.c1v_dec_head_fwd_check(C.byref(args)) != 0, kw
        assert text in lib.c1v_last_error(), (kw, lib.c1v_last_error())
    assert lib.c1v_dec_head_fwd_check(None) != 0
