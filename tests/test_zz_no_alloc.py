"""c1v_vocab_argmax_noalloc (static device scratch, no allocation) against c1v_vocab_argmax (caller scratch), through the
C ABI: identical (score, index) on two calls in a row, the second with a smaller input, so that stale partial results of
the first call would show."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _case(M, N, K, seed):
    from conv1d_text_vae_amd import _lib as L
    from conv1d_text_vae_amd import engine
    rng = np.random.RandomState(seed)
    W = rng.randn(N, K).astype(np.float32)
    W /= np.linalg.norm(W, axis=1, keepdims=True)
    W[N // 2] = W[3]                                            # an exact tie: the lower index must win in both paths
    q = W[rng.randint(0, N, size=M)] + 0.05 * rng.randn(M, K).astype(np.float32)
    q[0] = W[N // 2]
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    dec = engine.VocabDecoder(W, M, dtype='bf16', n_base=1000)
    dec.load_queries(torch.from_numpy(q))
    ref_s, ref_i = dec.run()                                    # existing entry point, caller-provided scratch
    ref_s, ref_i = ref_s.clone(), ref_i.clone()
    score = torch.full((M,), -7.0, dtype=torch.float32, device='cuda')
    idx = torch.full((M,), -7, dtype=torch.int32, device='cuda')
    a = L.VocabArgmaxNoallocArgs(dec.q.data_ptr(), M, dec.kp, dec.w.data_ptr(), N, dec.n_pad, 1000, score.data_ptr(),
                                 idx.data_ptr())
    return L, dec, a, score, idx, ref_s, ref_i


def test_noalloc_argmax_equals_the_allocating_entry_point_twice_in_a_row():
    lib = None
    for (M, N, K) in ((3000, 5000, 300), (777, 1300, 300)):     # the second call is smaller in both dimensions
        L, dec, a, score, idx, ref_s, ref_i = _case(M, N, K, seed=M)
        lib = L.load()
        free_before = torch.cuda.mem_get_info()[0]
        L.check(lib.c1v_vocab_argmax_noalloc(C.byref(a), _stream()), 'c1v_vocab_argmax_noalloc')
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before          # nothing was allocated on the device
        assert np.array_equal(idx.cpu().numpy(), ref_i.cpu().numpy())
        assert np.array_equal(score.cpu().numpy(), ref_s.cpu().numpy())
        assert int(idx[0]) == 1003
    # a narrow table (kp = 64) right after the wide ones
    L, dec, a, score, idx, ref_s, ref_i = _case(129, 200, 60, seed=5)
    L.check(lib.c1v_vocab_argmax_noalloc(C.byref(a), _stream()), 'c1v_vocab_argmax_noalloc')
    torch.cuda.synchronize()
    assert np.array_equal(idx.cpu().numpy(), ref_i.cpu().numpy()) and np.array_equal(score.cpu().numpy(), ref_s.cpu().numpy())


def test_noalloc_argmax_rejects_what_does_not_fit():
    from conv1d_text_vae_amd import _lib as L
    lib = L.load()
    t = torch.zeros(64, device='cuda')
    a = L.VocabArgmaxNoallocArgs(t.data_ptr(), L.NOALLOC_MAX_QUERIES + 1, 320, t.data_ptr(), 64, 64, 0, t.data_ptr(), t.data_ptr())
    assert lib.c1v_vocab_argmax_noalloc(C.byref(a), _stream()) != 0
    assert b'static scratch' in lib.c1v_last_error()
    a = L.VocabArgmaxNoallocArgs(t.data_ptr(), 10, 96, t.data_ptr(), 64, 64, 0, t.data_ptr(), t.data_ptr())
    assert lib.c1v_vocab_argmax_noalloc(C.byref(a), _stream()) != 0
    assert b"kp = 96" in lib.c1v_last_error()
