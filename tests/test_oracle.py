"""Pins the CPU oracle (oracle/vae_oracle.py).

1. a10-i is pinned by replaying the reference's own known-answer tests
   (reference tests/test_conv1d_text_vae.py:1544-1619).
2. a1-a9 have no golden values in the reference ("parity unpinned"); the NumPy restatement is
   cross-checked here against an independent torch-autograd restatement in float64 (forward, loss, every
   gradient) and the Nadam step against a literal transcription of the Keras-2.2 update rule.
3. find_best_words invariants (reference tests :1129-1152) on a synthetic embedding table.
"""
import math
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import vae_oracle as O


# ------------------------------------------------------------------ a10-i known-answer tests
def _source(end_idx, first_last=None):
    rng = np.random.RandomState(end_idx)
    src = np.hstack((rng.uniform(-1.0, 1.0, (20, 6)), rng.uniform(-3.0, -2.0, (20, 1))))
    src[end_idx] = 0.0
    src[end_idx][-1] = 1.0
    if first_last is not None:
        src[0][-1] = first_last
    true = np.zeros((20, 6))
    for i in range(end_idx):
        true[i] = src[i][:6]
        true[i] /= np.linalg.norm(true[i])
    return src, true


def test_postprocess_positive01():          # reference tests:1544-1573
    src, true = _source(11)
    out = O.postprocess_reconstructed_word_vectors(src.reshape(1, 20, 7))
    assert out.shape == (1, 20, 6)
    np.testing.assert_allclose(out[0], true, atol=1e-7)


def test_postprocess_positive02():          # reference tests:1575-1605 (EOS at step 0 -> end = max(0,1) = 1)
    src, true = _source(1, first_last=1.5)
    out = O.postprocess_reconstructed_word_vectors(src.reshape(1, 20, 7))
    np.testing.assert_allclose(out[0], true, atol=1e-7)


def test_postprocess_negative01():          # reference tests:1607-1619
    src, _ = _source(11)
    with pytest.raises(ValueError, match=re.escape(
            'The `word_vectors` parameter is wrong! Expected 3-D array, got 2-D array!')):
        O.postprocess_reconstructed_word_vectors(src)


# ------------------------------------------------------------------ geometry
def test_geometry_c2():
    g = O.VAEGeometry(32, 32, 302, 302, (128,), 3, 256)
    assert g.enc_len_conv == [30] and g.enc_len_pool == [15] and g.flat_size == 1920
    assert g.n_times == 15 and g.width == 18
    assert g.dec_len_deconv == [17] and g.dec_len_up == [34] and g.crop == 2
    assert g.n_params() == 1292348


def test_geometry_c4():
    g = O.VAEGeometry(128, 128, 302, 302, (128, 256, 256, 512, 512), 3, 256)
    assert g.enc_len_conv == [126, 61, 28, 12, 4] and g.enc_len_pool == [63, 30, 14, 6, 2]
    assert g.flat_size == 1024 and g.n_times == 3 and g.width == 86
    assert g.dec_len_up == [10, 24, 52, 108, 220] and g.crop == 92
    assert g.n_params() == 3907888


# ------------------------------------------------------------------ torch restatement
def _torch_forward(geom, P, Wv, x, ids, eps, mask_e, mask_d):
    k = geom.kernel_size
    h = x.permute(0, 2, 1)                                   # [B, C, L]
    for i in range(len(geom.n_filters)):
        W = P['encoder_conv1d_%d/kernel' % (i + 1)]          # [k, C, F]
        b = P['encoder_conv1d_%d/bias' % (i + 1)]
        h = F.relu(F.conv1d(h, W.permute(2, 1, 0), b))
        if geom.use_batch_norm:
            nm = 'encoder_batchnorm_%d' % (i + 1)
            h = F.batch_norm(h, None, None, P[nm + '/gamma'], P[nm + '/beta'], training=True, eps=O.BN_EPS)
        h = F.max_pool1d(h, 2)
    flat = h.permute(0, 2, 1).reshape(h.shape[0], -1)
    if mask_e is not None:
        flat = flat * mask_e / O.KEEP_PROB
    zm = flat @ P['z_mean/kernel'] + P['z_mean/bias']
    zl = flat @ P['z_log_var/kernel'] + P['z_log_var/bias']
    z = zm + torch.exp(zl) * eps
    zd = z * mask_d / O.KEEP_PROB if mask_d is not None else z
    d = F.relu(zd @ P['deconv_decoder_dense/kernel'] + P['deconv_decoder_dense/bias'])
    d = d.reshape(-1, geom.n_times, geom.width).permute(0, 2, 1)
    for i in range(len(geom.dec_filters)):
        Wt = P['deconv_decoder_%d_deconv1d_part2/kernel' % (i + 1)]      # [k, 1, F, C]
        b = P['deconv_decoder_%d_deconv1d_part2/bias' % (i + 1)]
        # torch conv_transpose1d weight: [C_in, C_out, k];  out[t] = sum_j x[t-j] w[:, :, j]
        d = F.relu(F.conv_transpose1d(d, Wt[:, 0].permute(2, 1, 0), b))
        if geom.use_batch_norm:
            nm = 'deconv_decoder_batchnorm_%d' % (i + 1)
            d = F.batch_norm(d, None, None, P[nm + '/gamma'], P[nm + '/beta'], training=True, eps=O.BN_EPS)
        d = d.repeat_interleave(2, dim=2)
    d = d[:, :, :geom.output_text_size]
    Wo = P['deconv_decoder_embeddings/kernel']
    pl, pr = (k - 1) // 2, k // 2
    y = F.conv1d(F.pad(d, (pl, pr)), Wo.permute(2, 1, 0), P['deconv_decoder_embeddings/bias']).permute(0, 2, 1)
    # loss
    yhat = y / torch.sqrt(torch.clamp((y * y).sum(-1, keepdim=True), min=O.L2_EPS))
    s = (yhat @ Wv.T) / O.TAU
    p = torch.softmax(s, dim=-1)
    p = p / p.sum(-1, keepdim=True)
    p = torch.clamp(p, O.XENT_EPS, 1 - O.XENT_EPS)
    B, T = ids.shape
    onehot = torch.zeros(B, T, Wv.shape[0], dtype=y.dtype)
    for b_ in range(B):
        for t_ in range(T):
            if ids[b_, t_] >= 0:
                onehot[b_, t_, ids[b_, t_]] = 1.0
    xent = -(onehot * torch.log(p)).sum(-1)
    mask = (onehot != 0).any(-1).to(y.dtype)
    xent = (xent * mask).sum(-1) / mask.sum(-1)
    kl = (1 + zl - zm * zm - torch.exp(zl)).mean(-1)
    return (xent - kl).mean(), zm, zl, y


CASES = [
    dict(B=4, T_in=12, T_out=11, E=8, n_filters=(6,), k=3, D=5, N=17),
    dict(B=3, T_in=30, T_out=26, E=6, n_filters=(5, 7, 4), k=3, D=9, N=23),
    dict(B=2, T_in=15, T_out=14, E=5, n_filters=(4, 6), k=4, D=3, N=11),
    dict(B=2, T_in=9, T_out=9, E=5, n_filters=(4,), k=2, D=3, N=11),
]


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('dropout', [False, True])
def test_forward_backward_matches_torch_autograd(case, dropout):
    V = case['E'] + 2
    geom = O.VAEGeometry(case['T_in'], case['T_out'], V, V, case['n_filters'], case['k'], case['D'])
    params = O.init_params(geom, seed=3, dtype=np.float64)
    rng = np.random.RandomState(0)
    for name in params:                                     # non-zero biases so their grads are exercised
        if name.endswith('/bias'):
            params[name] = rng.uniform(-0.1, 0.1, params[name].shape)
    Wv = O.synthetic_vocab(case['N'] - 2, case['E'], seed=5, dtype=np.float64)
    x, ids = O.synthetic_batch(Wv, Wv.shape[0], case['B'], case['T_in'], case['T_out'], seed=7)
    eps = rng.standard_normal((case['B'], case['D']))
    mask_e = (rng.uniform(size=(case['B'], geom.flat_size)) < 0.7) if dropout else None
    mask_d = (rng.uniform(size=(case['B'], case['D'])) < 0.7) if dropout else None
    loss, grads, acts = O.train_forward_backward(geom, params, Wv, x, ids, eps, mask_e, mask_d)

    P = {k_: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k_, v in params.items()}
    tl, zm, zl, y = _torch_forward(
        geom, P, torch.tensor(Wv), torch.tensor(x), ids, torch.tensor(eps),
        None if mask_e is None else torch.tensor(mask_e.astype(np.float64)),
        None if mask_d is None else torch.tensor(mask_d.astype(np.float64)))
    tl.backward()
    assert abs(loss - tl.item()) < 1e-10 * max(1.0, abs(loss))
    np.testing.assert_allclose(acts['z_mean'], zm.detach().numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(acts['y_pred'], y.detach().numpy(), rtol=1e-10, atol=1e-12)
    for name in params:
        np.testing.assert_allclose(grads[name], P[name].grad.numpy(), rtol=1e-8, atol=1e-12, err_msg=name)


@pytest.mark.parametrize('case', CASES[:3])
def test_batchnorm_forward_backward_matches_torch_autograd(case):
    """use_batch_norm=True (:1229-1230, :1259-1260): Keras BatchNormalization in its training phase = normalisation by
    the batch statistics over (B, L), which is torch's F.batch_norm(training=True); gradients of every weight, gamma and
    beta included, against autograd; moving averages against the Keras update rule; inference against the moving averages."""
    V = case['E'] + 2
    geom = O.VAEGeometry(case['T_in'], case['T_out'], V, V, case['n_filters'], case['k'], case['D'], use_batch_norm=True)
    params = O.init_params(geom, seed=3, dtype=np.float64)
    rng = np.random.RandomState(0)
    for name in params:
        if name.endswith(('/bias', '/beta')):
            params[name] = rng.uniform(-0.1, 0.1, params[name].shape)
        elif name.endswith('/gamma'):
            params[name] = rng.uniform(0.5, 1.5, params[name].shape) * rng.choice([-1.0, 1.0], params[name].shape)
        elif name.endswith('/moving_mean'):
            params[name] = rng.uniform(-0.2, 0.2, params[name].shape)
        elif name.endswith('/moving_variance'):
            params[name] = rng.uniform(0.5, 1.5, params[name].shape)
    Wv = O.synthetic_vocab(case['N'] - 2, case['E'], seed=5, dtype=np.float64)
    x, ids = O.synthetic_batch(Wv, Wv.shape[0], case['B'], case['T_in'], case['T_out'], seed=7)
    eps = rng.standard_normal((case['B'], case['D']))
    mask_e = rng.uniform(size=(case['B'], geom.flat_size)) < 0.7
    mask_d = rng.uniform(size=(case['B'], case['D'])) < 0.7
    loss, grads, acts = O.train_forward_backward(geom, params, Wv, x, ids, eps, mask_e, mask_d)
    trainable = [k_ for k_ in params if not k_.endswith(('/moving_mean', '/moving_variance'))]
    assert sorted(grads) == sorted(trainable)
    P = {k_: torch.tensor(v, dtype=torch.float64, requires_grad=k_ in trainable) for k_, v in params.items()}
    tl, zm, zl, y = _torch_forward(geom, P, torch.tensor(Wv), torch.tensor(x), ids, torch.tensor(eps),
                                   torch.tensor(mask_e.astype(np.float64)), torch.tensor(mask_d.astype(np.float64)))
    tl.backward()
    assert abs(loss - tl.item()) < 1e-10 * max(1.0, abs(loss))
    np.testing.assert_allclose(acts['y_pred'], y.detach().numpy(), rtol=1e-9, atol=1e-11)
    for name in trainable:
        np.testing.assert_allclose(grads[name], P[name].grad.numpy(), rtol=1e-7, atol=1e-11, err_msg=name)
    # moving averages: momentum 0.99 towards the batch mean / the batch variance times n / (n - (1 + eps))
    a0 = np.maximum(O.conv1d_valid(x, params['encoder_conv1d_1/kernel'], params['encoder_conv1d_1/bias']), 0)
    n = a0.shape[0] * a0.shape[1]
    want_mean = 0.99 * params['encoder_batchnorm_1/moving_mean'] + 0.01 * a0.mean(axis=(0, 1))
    want_var = 0.99 * params['encoder_batchnorm_1/moving_variance'] + 0.01 * a0.var(axis=(0, 1)) * n / (n - 1.001)
    np.testing.assert_allclose(acts['moving']['encoder_batchnorm_1/moving_mean'], want_mean, rtol=1e-12)
    np.testing.assert_allclose(acts['moving']['encoder_batchnorm_1/moving_variance'], want_var, rtol=1e-12)
    assert len(acts['moving']) == 2 * 2 * len(case['n_filters'])
    # inference normalises with the moving averages (no batch coupling: one sample alone gives the same row)
    zm_all, _ = O.encoder_forward(geom, params, x)
    zm_one, _ = O.encoder_forward(geom, params, x[:1])
    np.testing.assert_allclose(zm_all[:1], zm_one, rtol=1e-12, atol=1e-14)
    # and a Nadam step leaves the moving averages to the forward pass
    before = params['encoder_batchnorm_1/moving_mean'].copy()
    O.nadam_step(params, grads, O.NadamState())
    assert np.array_equal(before, params['encoder_batchnorm_1/moving_mean'])


def test_loss_clip_kills_gradient():
    """K.clip(p, 1e-7, 1-1e-7) has zero gradient outside the interval (tf.clip_by_value)."""
    V, N = 6, 9
    Wv = O.synthetic_vocab(N - 2, V - 2, seed=1, dtype=np.float64)
    y = np.zeros((1, 2, V))
    y[0, 0] = -Wv[3] * 5.0            # cos(target) = -1 -> p_t ~ e^-20/... < 1e-7 => clipped
    y[0, 1] = Wv[2] + 0.1
    ids = np.array([[3, 2]], dtype=np.int32)
    zm = np.zeros((1, 2))
    zl = np.zeros((1, 2))
    c = {}
    O.vae_loss(y, Wv, ids, zm, zl, cache=c)
    assert c['p_t'][0] < O.XENT_EPS
    dy, _, _ = O.vae_loss_bwd(y, Wv, ids, zm, zl, c)
    assert np.all(dy[0, 0] == 0) and np.any(dy[0, 1] != 0)


# ------------------------------------------------------------------ Nadam (Keras 2.2) literal transcription
def test_nadam_matches_keras_rule():
    rng = np.random.RandomState(1)
    p0 = {'a': rng.randn(7, 3), 'b': rng.randn(5)}
    params = {k_: v.copy() for k_, v in p0.items()}
    st = O.NadamState()
    # literal Keras-2.2 `Nadam.get_updates`
    it, msched = 0, 1.0
    ms = {k_: np.zeros_like(v) for k_, v in p0.items()}
    vs = {k_: np.zeros_like(v) for k_, v in p0.items()}
    ref = {k_: v.copy() for k_, v in p0.items()}
    for step in range(5):
        grads = {k_: rng.randn(*v.shape) * (30.0 if step == 2 else 1.0) for k_, v in p0.items()}
        norm = O.nadam_step(params, grads, st)
        n = math.sqrt(sum(float(np.sum(g * g)) for g in grads.values()))
        assert abs(norm - n) < 1e-9
        gs = {k_: (g * 10.0 / n if n >= 10.0 else g) for k_, g in grads.items()}
        t = it + 1
        mc_t = 0.9 * (1.0 - 0.5 * (0.96 ** (t * 0.004)))
        mc_t1 = 0.9 * (1.0 - 0.5 * (0.96 ** ((t + 1) * 0.004)))
        ms_new = msched * mc_t
        ms_next = msched * mc_t * mc_t1
        for k_ in ref:
            g = gs[k_]
            g_prime = g / (1.0 - ms_new)
            m_t = 0.9 * ms[k_] + 0.1 * g
            m_t_prime = m_t / (1.0 - ms_next)
            v_t = 0.999 * vs[k_] + 0.001 * g * g
            v_t_prime = v_t / (1.0 - 0.999 ** t)
            m_bar = (1.0 - mc_t) * g_prime + mc_t1 * m_t_prime
            ref[k_] = ref[k_] - 0.002 * m_bar / (np.sqrt(v_t_prime) + 1e-7)
            ms[k_], vs[k_] = m_t, v_t
        it, msched = t, ms_new
        for k_ in ref:
            np.testing.assert_allclose(params[k_], ref[k_], rtol=1e-12, atol=1e-14)


# ------------------------------------------------------------------ nearest word (reference tests :1129-1152)
def _table(E=12, Nw=40):
    rng = np.random.RandomState(2)
    raw = rng.randn(Nw, E).astype(np.float32)
    words = ['w%02d' % i for i in range(Nw)]
    return raw, raw / np.linalg.norm(raw, axis=1, keepdims=True), words


def test_find_best_words_eos_unknown_special():
    raw, nrm, words = _table()
    E = raw.shape[1]
    v = np.zeros((E + 2,), dtype=np.float32)
    v[E + 1] = 1.0
    assert O.find_best_words(v, nrm, words, raw, 3) is None                 # :1129-1133
    v = np.zeros((E + 2,), dtype=np.float32)
    v[E] = 1.0
    assert O.find_best_words(v, nrm, words, raw, 3) == []                   # :1135-1140
    v = np.zeros((E + 4,), dtype=np.float32)
    v[E + 1] = 1.0
    res = O.find_best_words(v, nrm, words, raw, 3, ('\t', '\n'))            # :1142-1152
    assert len(res) == 1 and res[0][0] == '\n' and abs(res[0][1]) < 1e-6


def test_find_best_words_topn_sorted():
    raw, nrm, words = _table()
    E = raw.shape[1]
    v = np.zeros((E + 2,), dtype=np.float32)
    v[:E] = raw[7]
    res = O.find_best_words(v, nrm, words, raw, 3)                           # :1114-1127
    assert len(res) == 3 and res[0][0] == 'w07'
    assert all(res[i][1] <= res[i + 1][1] for i in range(2)) and res[0][1] >= -1e-6


def test_vocab_argmax_first_index_wins_ties():
    Wn = np.eye(4, dtype=np.float32)[[0, 1, 1, 2]]
    q = np.array([[0, 1, 0, 0], [0, 0, 1, 0]], dtype=np.float32)
    s, i = O.vocab_argmax(q, Wn)
    assert i.tolist() == [1, 3] and s.tolist() == [1.0, 1.0]


# ------------------------------------------------------------------ RNG
def test_philox_known_answer():
    # Random123 known-answer vectors for philox4x32-10
    out = O.philox4x32(np.array([[0, 0, 0, 0]], dtype=np.uint32), (0, 0))
    assert [hex(v) for v in out[0]] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    out = O.philox4x32(np.array([[0xffffffff] * 4], dtype=np.uint32), (0xffffffff, 0xffffffff))
    assert [hex(v) for v in out[0]] == ['0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']
    out = O.philox4x32(np.array([[0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344]], dtype=np.uint32),
                       (0xa4093822, 0x299f31d0))
    assert [hex(v) for v in out[0]] == ['0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']


def test_philox_streams_are_offset_invariant():
    a = O.philox_normal(100, seed=9, stream=3)
    b = O.philox_normal(40, seed=9, stream=3, offset=60)
    assert np.array_equal(a[60:], b)
    m = O.philox_bernoulli(100000, 0.7, seed=9, stream=1)
    assert abs(m.mean() - 0.7) < 0.01
    assert abs(O.philox_normal(100000, 1, 2).std() - 1.0) < 0.02
