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
               'c1v_grad_finalize_args': _lib.GradFinalizeArgs}
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
