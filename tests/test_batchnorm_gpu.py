"""use_batch_norm=True on the GPU (reference :1229-1230, :1238-1239, :1259-1260, :1270-1272): the
BatchNormalization kernels around the conv GEMMs against the oracle, which itself is pinned to torch autograd
(tests/test_oracle.py::test_batchnorm_forward_backward_matches_torch_autograd)."""
import numpy as np
import pytest
import torch

from oracle import vae_oracle as O
from parity import assert_activation, rel_err, rel_l2, row_rel_l2

pytestmark = pytest.mark.gpu

CASES = {
    'bn1': dict(B=6, T_in=13, T_out=12, E=14, n_filters=(10,), k=3, D=5, N=40),
    'bn3': dict(B=5, T_in=30, T_out=27, E=30, n_filters=(12, 9, 8), k=3, D=7, N=70),
    'bn_k4': dict(B=4, T_in=17, T_out=15, E=6, n_filters=(5, 6), k=4, D=4, N=20),
    'bn_mid': dict(B=16, T_in=20, T_out=18, E=62, n_filters=(48, 40), k=3, D=24, N=300),
}


def make_case(c, seed=0):
    V = c['E'] + 2
    og = O.VAEGeometry(c['T_in'], c['T_out'], V, V, c['n_filters'], c['k'], c['D'], use_batch_norm=True)
    params = O.init_params(og, seed=3 + seed, dtype=np.float32)
    rng = np.random.RandomState(seed)
    for name in params:
        if name.endswith(('/bias', '/beta')):
            params[name] = rng.uniform(-0.1, 0.1, params[name].shape).astype(np.float32)
        elif name.endswith('/gamma'):
            params[name] = (rng.uniform(0.6, 1.4, params[name].shape) * rng.choice([-1.0, 1.0], params[name].shape)).astype(np.float32)
        elif name.endswith('/moving_mean'):
            params[name] = rng.uniform(0.0, 0.3, params[name].shape).astype(np.float32)
        elif name.endswith('/moving_variance'):
            params[name] = rng.uniform(0.5, 1.5, params[name].shape).astype(np.float32)
    Wv = O.synthetic_vocab(c['N'] - 2, c['E'], seed=5 + seed)
    x, ids = O.synthetic_batch(Wv, Wv.shape[0], c['B'], c['T_in'], c['T_out'], seed=7 + seed)
    eps = rng.standard_normal((c['B'], c['D'])).astype(np.float32)
    me = rng.uniform(size=(c['B'], og.flat_size)) < 0.7
    md = rng.uniform(size=(c['B'], c['D'])) < 0.7
    return og, params, Wv, x, ids, eps, me, md


def build(c, params, Wv, dtype, B=None, **kw):
    from conv1d_text_vae_amd import engine, geometry
    V = c['E'] + 2
    g = geometry.Geometry(c['T_in'], c['T_out'], V, V, c['n_filters'], c['k'], c['D'], use_batch_norm=True)
    return engine.VAEEngine(g, B or c['B'], Wv, dtype=dtype, weights=params, **kw)


def load_fixed(eng, x, ids, eps, me, md):
    eng.load_batch(x, ids)
    eng.eps.copy_(torch.from_numpy(eps))
    eng.keep_e.copy_(torch.from_numpy(me.astype(np.uint8)))
    eng.keep_d.copy_(torch.from_numpy(md.astype(np.uint8)))


@pytest.mark.parametrize('name', list(CASES))
@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_batchnorm_train_step_parity(name, dtype):
    c = CASES[name]
    og, params, Wv, x, ids, eps, me, md = make_case(c)
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    loss, grads, acts = O.train_forward_backward(og, p64, Wv.astype(np.float64), x.astype(np.float64), ids,
                                                 eps.astype(np.float64), me, md)
    eng = build(c, params, Wv, dtype)
    load_fixed(eng, x, ids, eps, me, md)
    eng._run(eng.train_ops(False))
    torch.cuda.synchronize()
    D = c['D']
    mlv = eng.mlv.cpu().numpy()
    for k_, got in (('z_mean', mlv[:, :D]), ('z_log_var', mlv[:, D:]), ('y_pred', eng.y_pred().cpu().numpy())):
        if dtype == 'f32':
            assert_activation(got, acts[k_], dtype, k_)
        else:
            # Measured exception to the 1e-2 bf16 bar: the normalisation shifts every element by
            # -mean / std, so the exact zeros a ReLU leaves (half of a conv output) become rounded non-zeros -- about twice
            # the rounding error of the un-normalised net per layer.  The flag is off by default; fp32 holds 1e-4.
            r2, rw = rel_l2(got, acts[k_]), row_rel_l2(got, acts[k_])
            print('bf16 + batch norm', name, k_, 'tensor rel-L2 %.3e  worst row %.3e' % (r2, rw))
            assert r2 < 2e-2 and rw < 8e-2, (k_, r2, rw)
    assert abs(eng.loss() - loss) < (1e-4 if dtype == 'f32' else 1e-2) * max(1.0, abs(loss)), (eng.loss(), loss)
    g = eng.get_grads()
    assert sorted(g) == sorted(grads)
    total = np.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for v in grads.values()))
    for k_, ref in grads.items():
        if dtype == 'f32':
            assert rel_err(g[k_], ref) < 5e-4, (k_, rel_err(g[k_], ref))
        else:
            # flipped ReLU / pool choices as without the normalisation; on top, the bias of a conv in front of a
            # normalisation has a gradient that nearly cancels per channel (a small difference of large sums), so its own
            # norm is no scale for its error: every tensor is also held against the norm of the whole gradient
            err = np.linalg.norm(g[k_].astype(np.float64) - ref)
            assert err < 0.3 * max(np.linalg.norm(ref), 1e-30) or err < 0.01 * total, (k_, err, np.linalg.norm(ref), total)
            assert err < 0.1 * total, (k_, err, total)
    # moving averages after one training forward (momentum 0.99, Keras' n / (n - (1 + eps)) variance correction)
    w = eng.get_weights()
    for k_, ref in acts['moving'].items():
        np.testing.assert_allclose(w[k_], ref, rtol=2e-5 if dtype == 'f32' else 2e-2, atol=1e-6 if dtype == 'f32' else 2e-3, err_msg=k_)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_batchnorm_inference_uses_the_moving_averages(dtype):
    c = CASES['bn3']
    og, params, Wv, x, ids, eps, me, md = make_case(c, seed=2)
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    zm, _ = O.encoder_forward(og, p64, x.astype(np.float64))
    y = O.decoder_forward(og, p64, zm)
    ref_eval = O.eval_loss(og, p64, Wv.astype(np.float64), x.astype(np.float64), ids, eps.astype(np.float64))
    eng = build(c, params, Wv, dtype)
    load_fixed(eng, x, ids, eps, me, md)
    assert_activation(eng.encode().cpu().numpy(), zm, dtype, 'transform')
    eng.reconstruct(postprocess=False)
    assert_activation(eng.y_pred().cpu().numpy(), y, dtype, 'reconstruct')
    eng.eval_step(use_graph=False, fixed_rng=True)
    assert abs(eng.loss() - ref_eval) < (1e-4 if dtype == 'f32' else 1e-2) * max(1.0, abs(ref_eval))
    # inference has no batch coupling: the first sample alone gives the same latent row
    e1 = build(c, params, Wv, dtype, B=1)
    e1.load_batch(x=x[:1])
    np.testing.assert_allclose(e1.encode().cpu().numpy(), eng.encode().cpu().numpy()[:1], rtol=1e-5, atol=1e-6)
    # and it leaves the moving averages alone
    w = eng.get_weights()
    for k_ in params:
        if 'moving' in k_:
            assert np.array_equal(w[k_], params[k_]), k_


def test_batchnorm_nadam_trajectory_and_graph_replay():
    """Four optimizer steps (fp32): losses, weights (gamma / beta included) and moving averages follow the oracle; the
    captured graph replays to the same bits as the eager launches."""
    c = CASES['bn_mid']
    og, params, Wv, x, ids, eps, me, md = make_case(c, seed=1)
    ref_p = {k: v.astype(np.float64) for k, v in params.items()}
    st = O.NadamState()
    ref_losses = []
    for _ in range(4):
        loss, grads, acts = O.train_forward_backward(og, ref_p, Wv.astype(np.float64), x.astype(np.float64), ids,
                                                     eps.astype(np.float64), me, md)
        O.nadam_step(ref_p, grads, st)
        ref_p.update(acts['moving'])
        ref_losses.append(loss)
    res = []
    for use_graph in (False, True):
        eng = build(c, params, Wv, 'f32')
        load_fixed(eng, x, ids, eps, me, md)
        losses = []
        for _ in range(4):
            eng.train_step(use_graph=use_graph, fixed_rng=True)
            losses.append(eng.loss())
        torch.cuda.synchronize()
        res.append((losses, eng.get_weights()))
    np.testing.assert_allclose(res[0][0], ref_losses, rtol=1e-3)
    w = res[0][1]
    for k_, ref in ref_p.items():
        if 'moving' in k_:
            np.testing.assert_allclose(w[k_], ref, rtol=1e-4, atol=1e-6, err_msg=k_)
        else:
            upd_ref = ref - params[k_].astype(np.float64)
            assert rel_err(w[k_].astype(np.float64) - params[k_], upd_ref) < 1e-2, (k_, rel_err(w[k_].astype(np.float64) - params[k_], upd_ref))
    assert res[0][0] == res[1][0]
    for k_ in w:
        assert np.array_equal(w[k_], res[1][1][k_]), k_


def test_batchnorm_two_ranks_use_global_statistics():
    """Data parallel: two half-batch engines whose reduction sums are added (the all-reduce) reproduce the single-engine
    step on the whole batch -- statistics over the GLOBAL batch, so results do not depend on the GPU count."""
    c = CASES['bn_mid']
    og, params, Wv, x, ids, eps, me, md = make_case(c, seed=4)
    full = build(c, params, Wv, 'f32')
    load_fixed(full, x, ids, eps, me, md)
    full._run(full.train_ops(False))
    torch.cuda.synchronize()
    h = c['B'] // 2
    ranks = []
    for r in range(2):
        e = build(c, params, Wv, 'f32', B=h, world_size=2, rank=r)
        sl = slice(r * h, (r + 1) * h)
        load_fixed(e, x[sl], ids[sl], eps[sl], me[sl], md[sl])
        ranks.append(e)
    # the two "processes" in lock step, op by op (both op lists have the same order); where a rank would all-reduce a sum
    # vector, the two ranks' vectors are added and written back to both
    flat0, flat1 = _flatten(ranks[0], ranks[0].train_ops(False)), _flatten(ranks[1], ranks[1].train_ops(False))
    assert len(flat0) == len(flat1)
    n_sync = 0
    for a, b in zip(flat0, flat1):
        ta, tb = getattr(a, 'bn_sync', None), getattr(b, 'bn_sync', None)
        if ta is not None:
            tot = ta + tb
            ta.copy_(tot)
            tb.copy_(tot)
            n_sync += 1
        else:
            a()
            b()
    assert n_sync == 2 * len(ranks[0].bn)          # one forward and one backward reduction per normalisation layer
    torch.cuda.synchronize()
    tot = ranks[0].grads + ranks[1].grads
    np.testing.assert_allclose(tot.cpu().numpy(), full.grads.cpu().numpy(), rtol=3e-4, atol=1e-5)
    assert abs((ranks[0].loss() + ranks[1].loss()) - full.loss()) < 1e-5 * abs(full.loss())
    w0, wf = ranks[0].get_weights(), full.get_weights()
    for k_ in wf:
        if 'moving' in k_:
            np.testing.assert_allclose(w0[k_], wf[k_], rtol=1e-5, atol=1e-7, err_msg=k_)


def _flatten(eng, ops):
    out = []
    for o in ops:
        if getattr(o, '__name__', '') == '_run_backward':
            for _, _, fused in eng.bwd_stages:
                out += list(fused)
        else:
            out.append(o)
    return out
