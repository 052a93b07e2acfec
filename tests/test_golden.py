"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py).

CPU half: the oracle still produces them (a change of the restated arithmetic cannot slip by).
GPU half: the HIP path, through the C ABI, reproduces them within the north-star tolerances
(fp32: 1e-4 relative on activations / latents / loss, 2e-4 on gradients; bf16: 2e-2 / 1e-2).
"""
import os

import numpy as np
import pytest
import torch

from oracle import vae_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {
    'tiny': dict(B=4, T_in=12, T_out=11, E=8, n_filters=(6,), k=3, D=5, N=17),
    'deep': dict(B=3, T_in=30, T_out=26, E=30, n_filters=(5, 7, 4), k=3, D=9, N=70),
    'even_k': dict(B=2, T_in=15, T_out=14, E=5, n_filters=(4, 6), k=4, D=3, N=11),
}


def load(name):
    with np.load(os.path.join(HERE, 'golden', 'vae_%s.npz' % name), allow_pickle=False) as z:
        return {k.replace('__', '/'): z[k] for k in z.files}


def group(g, prefix):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def geometry(c):
    V = c['E'] + 2
    return O.VAEGeometry(c['T_in'], c['T_out'], V, V, c['n_filters'], c['k'], c['D'])


@pytest.mark.parametrize('name', list(CASES))
def test_oracle_reproduces_golden(name):
    c, g = CASES[name], load(name)
    og = geometry(c)
    p64 = {k: v.astype(np.float64) for k, v in group(g, 'param/').items()}
    W, x, eps = (g['in/' + k].astype(np.float64) for k in ('W', 'x', 'eps'))
    ids, me, md = g['in/ids'], g['in/mask_e'].astype(bool), g['in/mask_d'].astype(bool)
    loss, grads, acts = O.train_forward_backward(og, p64, W, x, ids, eps, me, md)
    assert abs(loss - float(g['out/loss'])) < 1e-12
    for k in ('z_mean', 'z_log_var', 'z', 'y_pred', 'xent_b'):
        np.testing.assert_allclose(acts[k], g['out/' + k], rtol=1e-6, atol=1e-7)
    for k, v in group(g, 'grad/').items():
        np.testing.assert_allclose(grads[k], v, rtol=1e-5, atol=1e-9)
    assert abs(O.eval_loss(og, p64, W, x, ids, eps) - float(g['out/eval_loss'])) < 1e-12
    zm, _ = O.encoder_forward(og, p64, x)
    np.testing.assert_allclose(zm, g['out/transform'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(O.postprocess_reconstructed_word_vectors(g['out/reconstruct']), g['out/postprocessed'],
                               rtol=1e-6, atol=1e-7)
    st = O.NadamState()
    for step in range(len(g['nadam/losses'])):
        l, gr, _ = O.train_forward_backward(og, p64, W, x, ids, eps, me, md)
        norm = O.nadam_step(p64, gr, st)
        assert abs(l - g['nadam/losses'][step]) < 1e-12 and abs(norm - g['nadam/norms'][step]) < 1e-10
    for k in p64:
        np.testing.assert_allclose(p64[k], g['nadam/' + k], rtol=1e-6, atol=1e-7)


def test_philox_streams_golden():
    with np.load(os.path.join(HERE, 'golden', 'philox_streams.npz')) as z:
        np.testing.assert_array_equal(O.philox_normal(64, seed=1234, stream=1, offset=0), z['normal'])
        np.testing.assert_array_equal(O.philox_normal(16, seed=1234, stream=1, offset=37), z['normal_off'])
        np.testing.assert_array_equal(O.philox_bernoulli(256, 0.7, seed=1234, stream=2, offset=0), z['bernoulli'])


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('name', list(CASES))
@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_hip_path_reproduces_golden(name, dtype):
    from conv1d_text_vae_amd import engine, geometry as G
    c, g = CASES[name], load(name)
    V = c['E'] + 2
    geom = G.Geometry(c['T_in'], c['T_out'], V, V, c['n_filters'], c['k'], c['D'])
    params = group(g, 'param/')
    eng = engine.VAEEngine(geom, c['B'], g['in/W'], dtype=dtype, weights=params)

    def load_inputs():
        eng.load_batch(g['in/x'], g['in/ids'])
        eng.eps.copy_(torch.from_numpy(g['in/eps']))
        eng.keep_e.copy_(torch.from_numpy(g['in/mask_e']))
        eng.keep_d.copy_(torch.from_numpy(g['in/mask_d']))
    load_inputs()
    eng._run(eng.train_ops(False))
    torch.cuda.synchronize()
    from parity import assert_activation
    D = c['D']
    mlv = eng.mlv.cpu().numpy()
    assert_activation(mlv[:, :D], g['out/z_mean'], dtype, 'z_mean')
    assert_activation(mlv[:, D:], g['out/z_log_var'], dtype, 'z_log_var')
    assert_activation(eng.z_out.cpu().numpy(), g['out/z'], dtype, 'z')
    assert_activation(eng.y_pred().cpu().numpy(), g['out/y_pred'], dtype, 'y_pred')
    loss = float(g['out/loss'])
    assert abs(eng.loss() - loss) < (1e-4 if dtype == 'f32' else 1e-2) * max(1.0, abs(loss))
    if dtype == 'f32':
        got = eng.get_grads()
        for k, v in group(g, 'grad/').items():
            assert rel_err(got[k], v) < 2e-4, (k, rel_err(got[k], v))
    # inference rows: transform() = z_mean without dropout; predict()'s decoder(encoder(x)) and EOS truncation
    assert_activation(eng.encode().cpu().numpy(), g['out/transform'], dtype, 'transform')
    eng.reconstruct(postprocess=True)
    y = eng.y_pred().cpu().numpy()
    assert_activation(y, g['out/reconstruct'], dtype, 'reconstruct')
    np.testing.assert_allclose(eng.post.cpu().numpy(), O.postprocess_reconstructed_word_vectors(y), rtol=1e-5, atol=1e-6)
    eng.eval_step(use_graph=False, fixed_rng=True)
    ev = float(g['out/eval_loss'])
    assert abs(eng.loss() - ev) < (1e-4 if dtype == 'f32' else 1e-2) * max(1.0, abs(ev))
    # optimizer trajectory
    losses = []
    for _ in range(len(g['nadam/losses'])):
        eng.train_step(use_graph=False, fixed_rng=True)
        losses.append(eng.loss())
    np.testing.assert_allclose(losses, g['nadam/losses'], rtol=1e-3 if dtype == 'f32' else 1e-2)
    if dtype == 'f32':
        w = eng.get_weights()
        for k, v0 in params.items():
            upd_ref = g['nadam/' + k].astype(np.float64) - v0
            upd = w[k].astype(np.float64) - v0
            assert rel_err(upd, upd_ref) < 5e-3, (k, rel_err(upd, upd_ref))
