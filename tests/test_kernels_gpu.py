"""GPU tests of the individual C-ABI ops that the engine test does not pin on its own: nearest-word arg-max (a10-ii),
EOS truncation (a10-i, against the reference's known-answer construction), Philox streams, and full-size
(BASELINE.json configs[1], configs[2]) runs checked through the oracle or through size-independent properties."""
import ctypes as C
import re

import numpy as np
import pytest
import torch

from oracle import vae_oracle as O

pytestmark = pytest.mark.gpu


def _mods():
    from conv1d_text_vae_amd import _lib, engine, geometry, synthetic
    return _lib, engine, geometry, synthetic


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ------------------------------------------------------------------------------------------------ a10-ii
@pytest.mark.parametrize('dtype,K,N,M', [('bf16', 300, 5000, 700), ('bf16', 302, 333, 129), ('bf16', 30, 1000, 64),
                                         ('bf16', 60, 130, 10), ('f32', 300, 3000, 200), ('bf16', 100, 900, 77),
                                         ('bf16', 150, 1100, 130), ('bf16', 200, 2100, 260), ('bf16', 256, 700, 33),
                                         ('bf16', 340, 600, 50)])
def test_vocab_argmax_matches_oracle(dtype, K, N, M):
    L, engine, _, _ = _mods()
    rng = np.random.RandomState(K + N)
    W = rng.randn(N, K).astype(np.float32)
    W /= np.linalg.norm(W, axis=1, keepdims=True)
    q = W[rng.randint(0, N, size=M)] + 0.05 * rng.randn(M, K).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    if dtype == 'bf16':       # compare on the values the kernel sees
        W = torch.from_numpy(W).to(torch.bfloat16).float().numpy()
        q = torch.from_numpy(q).to(torch.bfloat16).float().numpy()
    dec = engine.VocabDecoder(W, M, dtype=dtype)
    dec.load_queries(torch.from_numpy(q))
    score, idx = dec.run()
    torch.cuda.synchronize()
    s64 = q.astype(np.float64) @ W.astype(np.float64).T
    ref_idx = s64.argmax(axis=1)
    got = idx.cpu().numpy()
    # decoded ids are bit-exact except on documented near-ties: accept a different index only if its exact score is
    # within fp32 accumulation noise of the maximum
    bad = got != ref_idx
    if bad.any():
        gap = s64[np.arange(M), ref_idx][bad] - s64[np.arange(M), got][bad]
        assert np.all(gap < 2e-6), gap.max()
    np.testing.assert_allclose(score.cpu().numpy(), s64.max(axis=1), rtol=0, atol=2e-5)


def test_vocab_argmax_ties_pick_lowest_index_and_n_base():
    L, engine, _, _ = _mods()
    K, N = 64, 300
    W = np.zeros((N, K), dtype=np.float32)
    W[np.arange(N), np.arange(N) % K] = 1.0          # rows r and r+64k are identical
    q = np.eye(K, dtype=np.float32)[[5, 63, 0]]
    dec = engine.VocabDecoder(W, 3, dtype='bf16', n_base=1000)
    dec.load_queries(torch.from_numpy(q))
    score, idx = dec.run()
    assert idx.cpu().tolist() == [1005, 1063, 1000] and score.cpu().tolist() == [1.0, 1.0, 1.0]


def test_vocab_argmax_c3_shape_self_retrieval():
    """BASELINE configs[2] shape (M = 8192*64 rows, 200 000-row vocabulary, emb 300): every query is a vocabulary row,
    so the fused GEMM + arg-max must return exactly that row (size-independent property; the oracle would need 63 TFLOP)."""
    L, engine, _, synthetic = _mods()
    N, K, M = 200000, 302, 8192 * 64
    W = synthetic.synthetic_vocab(N - 2, K - 2, seed=3)
    Wb = torch.from_numpy(W).to(torch.bfloat16)
    # drop duplicates that bf16 rounding could create: not expected for random 300-d rows
    rng = np.random.RandomState(0)
    pick = rng.randint(0, N, size=M)
    dec = engine.VocabDecoder(W, M, dtype='bf16')
    dec.load_queries(Wb[torch.from_numpy(pick)].float())
    score, idx = dec.run()
    torch.cuda.synchronize()
    assert np.array_equal(idx.cpu().numpy(), pick)
    assert float(score.min()) > 0.99


# ------------------------------------------------------------------------------------------------ a10-i
def test_eos_truncate_known_answers():
    """The construction of reference tests/test_conv1d_text_vae.py:1544-1605, batched, through the HIP kernel."""
    L, _, _, _ = _mods()
    lib = L.load()
    rng = np.random.RandomState(1)
    B, T, V = 7, 20, 7
    src = np.concatenate([rng.uniform(-1, 1, (B, T, V - 1)), rng.uniform(-3, -2, (B, T, 1))], axis=2).astype(np.float32)
    ends = [11, 1, 0, 19, 5, T, 3]          # T = no EOS step at all
    true = np.zeros((B, T, V - 1), dtype=np.float32)
    for b, e in enumerate(ends):
        if e < T:
            src[b, e] = 0.0
            src[b, e, V - 1] = 1.0
        if b == 1:
            src[b, 0, V - 1] = 1.5          # EOS wins at step 0 -> end = max(0, 1) = 1   (:1582-1585)
        stop = max(e, 1) if e < T else T
        if b == 1:
            stop = 1
        true[b, :stop] = src[b, :stop, :V - 1] / np.linalg.norm(src[b, :stop, :V - 1], axis=-1, keepdims=True)
    y = torch.from_numpy(src).cuda()
    out = torch.zeros(B, T, V - 1, device='cuda')
    end = torch.zeros(B, dtype=torch.int32, device='cuda')
    L.check(lib.c1v_eos_truncate_normalize(y.data_ptr(), B, T, V, out.data_ptr(), end.data_ptr(), _stream()))
    np.testing.assert_allclose(out.cpu().numpy(), true, atol=1e-6)
    np.testing.assert_allclose(out.cpu().numpy(), O.postprocess_reconstructed_word_vectors(src), atol=1e-6)
    assert np.array_equal(end.cpu().numpy(), O.eos_end_index(src))
    # the same rows lying `ldy` values apart (the engine pads the decoder output's rows to 32 bytes): bit-identical
    ldy = 16
    wide = np.full((B, T, ldy), 7.0, dtype=np.float32)          # (the padding holds values that would win the arg-max)
    wide[:, :, :V] = src
    y2 = torch.from_numpy(wide).cuda()
    out2 = torch.zeros(B, T, V - 1, device='cuda')
    end2 = torch.zeros(B, dtype=torch.int32, device='cuda')
    L.check(lib.c1v_eos_truncate_normalize_ld(y2.data_ptr(), ldy, B, T, V, out2.data_ptr(), end2.data_ptr(), _stream()))
    assert np.array_equal(out2.cpu().numpy(), out.cpu().numpy(), equal_nan=True)
    assert np.array_equal(end2.cpu().numpy(), end.cpu().numpy())
    assert lib.c1v_eos_truncate_normalize_ld(y2.data_ptr(), V - 1, B, T, V, out2.data_ptr(), end2.data_ptr(), _stream()) != 0


# ------------------------------------------------------------------------------------------------ RNG
def test_philox_streams_match_oracle():
    L, _, _, _ = _mods()
    lib = L.load()
    n = 10007
    out = torch.zeros(n, device='cuda')
    L.check(lib.c1v_philox_normal(out.data_ptr(), n, 99, 5, 13, None, 0, _stream()))
    np.testing.assert_allclose(out.cpu().numpy(), O.philox_normal(n, seed=99, stream=5, offset=13), rtol=2e-4, atol=2e-5)
    m = torch.zeros(n, dtype=torch.uint8, device='cuda')
    L.check(lib.c1v_philox_bernoulli(m.data_ptr(), n, 0.7, 99, 6, 2, None, 0, _stream()))
    assert np.array_equal(m.cpu().numpy(), O.philox_bernoulli(n, 0.7, seed=99, stream=6, offset=2))


# ------------------------------------------------------------------------------------------------ full size (C2)
@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_c2_full_size_step_matches_oracle(dtype):
    """BASELINE configs[1]: batch 256, seq_len 32, emb 300, latent 256, vocabulary 8192 -- one train step against the
    oracle (float32 oracle, a few seconds), then 3 more graph-replayed steps must keep the loss finite and decreasing."""
    L, engine, geometry, synthetic = _mods()
    B, T, E, D, N = 256, 32, 300, 256, 8192
    V = E + 2
    W = synthetic.synthetic_vocab(N - 2, E)
    x, ids = synthetic.synthetic_batch(W, N, B, T, T)
    og = O.VAEGeometry(T, T, V, V, (128,), 3, D)
    params = O.init_params(og, seed=0)
    rng = np.random.RandomState(0)
    eps = rng.standard_normal((B, D)).astype(np.float32)
    me = rng.uniform(size=(B, og.flat_size)) < 0.7
    md = rng.uniform(size=(B, D)) < 0.7
    loss, grads, acts = O.train_forward_backward(og, params, W, x, ids, eps, me, md)
    g = geometry.Geometry(T, T, V, V, (128,), 3, D)
    eng = engine.VAEEngine(g, B, W, dtype=dtype, weights=params)
    eng.load_batch(x, ids)
    eng.eps.copy_(torch.from_numpy(eps))
    eng.keep_e.copy_(torch.from_numpy(me.astype(np.uint8)))
    eng.keep_d.copy_(torch.from_numpy(md.astype(np.uint8)))
    eng._run(eng.train_ops(False))
    torch.cuda.synchronize()
    from parity import assert_activation, engine_decisions, rel_l2
    if dtype == 'f32':
        # (the oracle itself runs in float32 here: 2e-4 instead of the 1e-4 the float64-oracle tests hold the kernels to)
        y = eng.y_pred().cpu().numpy()
        assert np.abs(y - acts['y_pred']).max() / np.abs(acts['y_pred']).max() < 2e-4
    else:
        mlv = eng.mlv.cpu().numpy()
        for k_, got in (('z_mean', mlv[:, :D]), ('z_log_var', mlv[:, D:]), ('z', eng.z_out.cpu().numpy()),
                        ('y_pred', eng.y_pred().cpu().numpy())):
            r, e = assert_activation(got, acts[k_], dtype, k_)
            print('C2 bf16', k_, 'worst row rel-L2 %.3e  max-rel %.3e' % (r, e))
    assert abs(eng.loss() - loss) < (2e-4 if dtype == 'f32' else 1e-2) * abs(loss)
    gg = eng.get_grads()
    if dtype == 'bf16':
        # the oracle's backward pass on the ReLU / max-pool choices the bf16 run made
        _, same_choice, _ = O.train_forward_backward(og, params, W, x, ids, eps, me, md, decisions=engine_decisions(eng))
    for k_, ref in grads.items():
        err = rel_l2(gg[k_], ref)
        if dtype == 'f32':
            assert err < 1e-3, (k_, err)
            continue
        # bf16, measured exception: a bf16 forward pass flips the ReLU / max-pool choice of the ~0.3 % of elements that are
        # within its rounding of a tie; a fraction f of flipped choices costs sqrt(f) of a gradient's Frobenius norm per
        # layer crossed (0.2 % after the loss, ~2 % after one ReLU layer, 5-7 % at the encoder).  With the choices held
        # equal the gradients agree to the activation tolerance.
        errs = rel_l2(gg[k_], same_choice[k_])
        print('C2 bf16 grad', k_, 'rel-L2 %.3e (same ReLU / pool choices: %.3e)' % (err, errs))
        assert err < 0.12, (k_, err)
        assert errs < 1e-2, (k_, errs)
    losses = []
    for _ in range(4):
        eng.train_step(use_graph=True, fixed_rng=True)
        losses.append(eng.loss())
    assert np.all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_c2_size_loss_curve_100_steps_bf16_vs_f32_within_one_percent():
    """north_star: loss curves within 1 % over 100 steps with fixed eps -- at the full C2 size (batch 256, seq_len 32,
    emb 300, latent 256, vocabulary 8192).  The oracle affords three steps at this size, so the fp32 HIP path is pinned to
    it there (float32 oracle) and the bf16 path is then held to the fp32 path over all 100 steps: same weights, the same
    eight batches cycled, the same eps and dropout masks."""
    L, engine, geometry, synthetic = _mods()
    B, T, E, D, N = 256, 32, 300, 256, 8192
    V = E + 2
    W = synthetic.synthetic_vocab(N - 2, E)
    og = O.VAEGeometry(T, T, V, V, (128,), 3, D)
    params = O.init_params(og, seed=0)
    rng = np.random.RandomState(7)
    batches = []
    for i in range(8):
        x, ids = synthetic.synthetic_batch(W, N, B, T, T, seed=100 + i)
        batches.append((x, ids, rng.standard_normal((B, D)).astype(np.float32),
                        rng.uniform(size=(B, og.flat_size)) < 0.7, rng.uniform(size=(B, D)) < 0.7))
    ref_p = {k: v.copy() for k, v in params.items()}
    st = O.NadamState()
    ref = []
    for step in range(3):
        x, ids, eps, me, md = batches[step % 8]
        loss, grads, _ = O.train_forward_backward(og, ref_p, W, x, ids, eps, me, md)
        O.nadam_step(ref_p, grads, st)
        ref.append(loss)
    g = geometry.Geometry(T, T, V, V, (128,), 3, D)
    curves = {}
    for dtype in ('f32', 'bf16'):
        eng = engine.VAEEngine(g, B, W, dtype=dtype, weights=params)
        dev_batches = [(torch.from_numpy(x).to(eng.dev), torch.from_numpy(ids).to(eng.dev), torch.from_numpy(eps).to(eng.dev),
                        torch.from_numpy(me.astype(np.uint8)).to(eng.dev), torch.from_numpy(md.astype(np.uint8)).to(eng.dev))
                       for x, ids, eps, me, md in batches]
        got = []
        for step in range(100):
            x, ids, eps, me, md = dev_batches[step % 8]
            eng.x_in.copy_(x)
            eng.ids_in.copy_(ids)
            eng.eps.copy_(eps)
            eng.keep_e.copy_(me)
            eng.keep_d.copy_(md)
            eng.train_step(use_graph=True, fixed_rng=True)
            got.append(eng.loss())
        curves[dtype] = np.array(got)
        del eng
    np.testing.assert_allclose(curves['f32'][:3], ref, rtol=1e-3)
    assert curves['f32'][-1] < 0.97 * curves['f32'][0]              # it trains
    dev = np.max(np.abs(curves['bf16'] - curves['f32']) / curves['f32'])
    print('C2 size: max relative deviation of the bf16 loss curve from the fp32 one over 100 steps: %.3e (loss %.3f -> %.3f)'
          % (dev, curves['f32'][0], curves['f32'][-1]))
    assert dev < 1e-2


def test_loss_curve_100_steps_within_one_percent():
    """north_star: loss curves within 1 % over 100 steps with fixed eps (and fixed dropout masks): oracle (float64) vs the
    fp32 HIP path vs the bf16 HIP path; same weights, same sequence of batches (10 batches cycled, as an epoch loop
    does), Nadam(clipnorm=10)."""
    L, engine, geometry, synthetic = _mods()
    c = dict(B=128, T_in=20, T_out=18, E=62, n_filters=(48,), k=3, D=24, N=3000)
    V = c['E'] + 2
    og = O.VAEGeometry(c['T_in'], c['T_out'], V, V, c['n_filters'], c['k'], c['D'])
    params = O.init_params(og, seed=1)
    Wv = O.synthetic_vocab(c['N'] - 2, c['E'], seed=2)
    rng = np.random.RandomState(4)
    batches = []
    for i in range(10):
        x, ids = O.synthetic_batch(Wv, Wv.shape[0], c['B'], c['T_in'], c['T_out'], seed=30 + i)
        eps = rng.standard_normal((c['B'], c['D'])).astype(np.float32)
        me = rng.uniform(size=(c['B'], og.flat_size)) < 0.7
        md = rng.uniform(size=(c['B'], c['D'])) < 0.7
        batches.append((x, ids, eps, me, md))
    ref_p = {k: v.astype(np.float64) for k, v in params.items()}
    st = O.NadamState()
    ref = []
    for step in range(100):
        x, ids, eps, me, md = batches[step % 10]
        loss, grads, _ = O.train_forward_backward(og, ref_p, Wv.astype(np.float64), x.astype(np.float64), ids,
                                                  eps.astype(np.float64), me, md)
        O.nadam_step(ref_p, grads, st)
        ref.append(loss)
    assert ref[-1] < 0.9 * ref[0]                      # it actually trains
    g = geometry.Geometry(c['T_in'], c['T_out'], V, V, c['n_filters'], c['k'], c['D'])
    for dtype in ('f32', 'bf16'):
        eng = engine.VAEEngine(g, c['B'], Wv, dtype=dtype, weights=params)
        got = []
        for step in range(100):
            x, ids, eps, me, md = batches[step % 10]
            eng.load_batch(x, ids)
            eng.eps.copy_(torch.from_numpy(eps))
            eng.keep_e.copy_(torch.from_numpy(me.astype(np.uint8)))
            eng.keep_d.copy_(torch.from_numpy(md.astype(np.uint8)))
            eng.train_step(use_graph=True, fixed_rng=True)
            got.append(eng.loss())
        dev = np.max(np.abs(np.array(got) - np.array(ref)) / np.array(ref))
        print(dtype, 'max relative deviation of the loss curve over 100 steps: %.3e (loss %.3f -> %.3f)' % (dev, ref[0], ref[-1]))
        np.testing.assert_allclose(got, ref, rtol=1e-2, err_msg=dtype)


@pytest.mark.gpu
@pytest.mark.parametrize('M', [1025, 4100, 5000, 8192, 16384, 16385])
def test_live_row_table_from_several_workgroups_matches_the_single_workgroup(M):
    """The step prologue builds the live-row table with up to 16 workgroups of 1 024 rows, each counting the rows in front
    of its range itself (live_rows_mb_body); c1v_live_rows builds it with one workgroup.  Positions and the control block
    must be the same integers, and the positions those of a NumPy cumulative count -- ragged ends (M not a multiple of 4 or
    of 1 024), all-dead and all-live stretches, and one size beyond the multi-workgroup limit (the prologue then falls back)."""
    from conv1d_text_vae_amd import _lib as L
    lib = L.load()
    dev = torch.device('cuda:0')
    rng = np.random.RandomState(M)
    ids = rng.randint(-1, 5, size=M).astype(np.int32)
    ids[ids > 0] = rng.randint(0, 1000, size=int((ids > 0).sum()))
    ids[100:700] = -1
    ids[2000:min(M, 3300)] = 7
    t_ids = torch.from_numpy(ids).to(dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    items, vocab_tiles, max_splits = max(256, -(-M // 128)), 128, 16
    res = []
    live = ids >= 0
    n_live = int(live.sum())
    for which in ('prologue', 'table', 'single'):
        pos = torch.full((M + 8,), -7, dtype=torch.int32, device=dev)
        ctl = torch.zeros(8, dtype=torch.int32, device=dev)
        rows = torch.full((M + 8,), -7, dtype=torch.int32, device=dev)
        cids = torch.full((M + 8,), -7, dtype=torch.int32, device=dev)
        la = L.LiveRowsArgs(t_ids.data_ptr(), M, items, vocab_tiles, max_splits, pos.data_ptr(), ctl.data_ptr(),
                            rows.data_ptr(), cids.data_ptr())
        if which == 'prologue':
            a = L.StepPrologueArgs()
            a.live = C.pointer(la)
            L.check(lib.c1v_step_prologue(C.byref(a), st), 'c1v_step_prologue')
        elif which == 'table':
            L.check(lib.c1v_live_rows_table(C.byref(la), st), 'c1v_live_rows_table')
        else:
            L.check(lib.c1v_live_rows(t_ids.data_ptr(), M, items, vocab_tiles, max_splits, pos.data_ptr(), ctl.data_ptr(), st), 'c1v_live_rows')
        torch.cuda.synchronize()
        if which != 'single':
            # the inverse table: compact row p -> (sample row, target id); nothing written behind the last live row
            np.testing.assert_array_equal(rows.cpu().numpy()[:n_live], np.nonzero(live)[0].astype(np.int32))
            np.testing.assert_array_equal(cids.cpu().numpy()[:n_live], ids[live])
            assert np.all(rows.cpu().numpy()[n_live:] == -7) and np.all(cids.cpu().numpy()[n_live:] == -7)
        res.append((pos.cpu().numpy(), ctl.cpu().numpy()))
    for k in (0, 1):
        np.testing.assert_array_equal(res[k][0], res[2][0])
        np.testing.assert_array_equal(res[k][1], res[2][1])
    want = np.where(live, np.cumsum(live) - 1, -1).astype(np.int32)
    np.testing.assert_array_equal(res[0][0][:M], want)
    assert np.all(res[0][0][M:] == -7) and res[0][1][0] == n_live


@pytest.mark.gpu
@pytest.mark.parametrize('M', [3000, 4096, 70000, 270336])
def test_live_row_table_of_the_large_batch_launch(M):
    """c1v_live_rows_gather (large batches: live-row table + input gather in one launch of 1 024-thread workgroups) builds
    the table -- positions, control block and the inverse table -- with one workgroup per 4 096 rows; against c1v_live_rows
    and NumPy.  The gather half of the launch is given a tiny job of its own and checked too."""
    from conv1d_text_vae_amd import _lib as L
    lib = L.load()
    dev = torch.device('cuda:0')
    rng = np.random.RandomState(M % 9973)
    ids = rng.randint(-3, 1000, size=M).astype(np.int32)
    ids[ids < 0] = -1
    ids[M // 3: M // 3 + 5000] = -1
    t_ids = torch.from_numpy(ids).to(dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    items, vocab_tiles, max_splits = max(256, -(-M // 128)) * 2, 512, 8
    live = ids >= 0
    n_live = int(live.sum())
    # gather job: 4 samples x 6 steps of a 40-column table into a bf16 operand with 64-channel rows
    Bg, Tg, Cg, cp = 4, 6, 40, 64
    table = torch.from_numpy(rng.randn(50, Cg).astype(np.float32)).to(dev)
    gid = torch.from_numpy(rng.randint(-1, 50, size=(Bg, Tg)).astype(np.int32)).to(dev)
    dst = torch.zeros(Bg * 8 * cp + 4096, dtype=torch.bfloat16, device=dev)
    ga = L.GatherInputArgs(table.data_ptr(), Cg, gid.data_ptr(), Tg, Cg, L.Seq(dst.data_ptr(), L.BF16, Bg, 8 * cp, 1, Tg, cp, Cg))
    pos = torch.full((M + 8,), -7, dtype=torch.int32, device=dev)
    ctl = torch.zeros(8, dtype=torch.int32, device=dev)
    rows = torch.full((M + 8,), -7, dtype=torch.int32, device=dev)
    cids = torch.full((M + 8,), -7, dtype=torch.int32, device=dev)
    la = L.LiveRowsArgs(t_ids.data_ptr(), M, items, vocab_tiles, max_splits, pos.data_ptr(), ctl.data_ptr(), rows.data_ptr(), cids.data_ptr())
    L.check(lib.c1v_live_rows_gather(C.byref(la), C.byref(ga), st), 'c1v_live_rows_gather')
    pos1 = torch.full((M + 8,), -7, dtype=torch.int32, device=dev)
    ctl1 = torch.zeros(8, dtype=torch.int32, device=dev)
    L.check(lib.c1v_live_rows(t_ids.data_ptr(), M, items, vocab_tiles, max_splits, pos1.data_ptr(), ctl1.data_ptr(), st), 'c1v_live_rows')
    torch.cuda.synchronize()
    np.testing.assert_array_equal(pos.cpu().numpy(), pos1.cpu().numpy())
    np.testing.assert_array_equal(ctl.cpu().numpy(), ctl1.cpu().numpy())
    np.testing.assert_array_equal(pos.cpu().numpy()[:M], np.where(live, np.cumsum(live) - 1, -1).astype(np.int32))
    np.testing.assert_array_equal(rows.cpu().numpy()[:n_live], np.nonzero(live)[0].astype(np.int32))
    np.testing.assert_array_equal(cids.cpu().numpy()[:n_live], ids[live])
    assert np.all(rows.cpu().numpy()[n_live:] == -7) and int(ctl.cpu().numpy()[0]) == n_live
    got = dst[:Bg * 8 * cp].float().cpu().numpy().reshape(Bg, 8, cp)[:, 1:1 + Tg, :Cg]
    want = np.where(gid.cpu().numpy()[:, :, None] >= 0, table.cpu().numpy()[np.maximum(gid.cpu().numpy(), 0)], 0.0)
    np.testing.assert_allclose(got, want, rtol=2 ** -8, atol=0)


@pytest.mark.gpu
def test_recorded_list_stops_at_a_failing_entry_and_stays_ordered():
    """c1v_run_list / c1v_run_list_fork (the recorded step; no counterpart in the reference): the first failing entry ends the
    list with its return code and its error text, later entries are not issued, and what was already issued on the second
    stream is joined, so that synchronising the main stream covers it."""
    from conv1d_text_vae_amd import _lib as L
    lib = L.load()
    dev = torch.device('cuda:0')
    B, D = 8, 16
    g = torch.Generator(device='cpu').manual_seed(5)
    mlv = torch.randn(B, 2 * D, generator=g).to(dev)
    eps = torch.randn(B, D, generator=g).to(dev)
    zd = torch.zeros(B, 64, device=dev)
    z = [torch.full((B, D), -7.0, device=dev) for _ in range(3)]

    def latent(rows, z_out):
        return L.LatentArgs(mlv.data_ptr(), rows, D, eps.data_ptr(), None, 0.7, 1.0, zd.data_ptr(), L.F32, 64, z_out.data_ptr(), None,
                            None, B, None, 0, 1, 0)
    items = [latent(B, z[0]), latent(0, z[1]), latent(B, z[2])]          # the second one fails its shape check
    n = len(items)
    fns = (C.c_void_p * n)(*[C.cast(lib.c1v_latent_head_fwd, C.c_void_p)] * n)
    args = (C.c_void_p * n)(*[C.cast(C.pointer(a), C.c_void_p) for a in items])
    want = (mlv[:, :D] + torch.exp(mlv[:, D:]) * eps).cpu().numpy()
    main = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    side = torch.cuda.Stream(dev)
    for forked in (False, True):
        for t in z:
            t.fill_(-7.0)
        torch.cuda.synchronize()
        if forked:
            flags = (C.c_int32 * n)(L.LIST_SIDE | L.LIST_FORK, 0, 0)
            rc = lib.c1v_run_list_fork(n, fns, args, flags, main, C.c_void_p(side.cuda_stream))
        else:
            rc = lib.c1v_run_list(n, fns, args, main)
        assert rc != 0 and b'latent_head_fwd' in lib.c1v_last_error()
        torch.cuda.current_stream(dev).synchronize()              # (the main stream only: the side entry must have been joined)
        np.testing.assert_allclose(z[0].cpu().numpy(), want, rtol=1e-6, atol=1e-6)
        assert float(z[1].min()) == -7.0 and float(z[2].max()) == -7.0
    # and a complete list returns 0
    ok = (C.c_void_p * 2)(args[0], args[2])
    assert lib.c1v_run_list(2, fns, ok, main) == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(z[2].cpu().numpy(), want, rtol=1e-6, atol=1e-6)
