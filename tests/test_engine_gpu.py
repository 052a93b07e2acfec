"""GPU parity of the HIP hot path against the CPU oracle.  Everything here goes through the C ABI
(libc1dvae.so via ctypes); the oracle is only the checker.

Tolerances (north_star): forward activations / latents within 1e-4 relative in fp32 and 1e-2 in bf16; gradients are
held to the same bars relative to the gradient's scale; the Nadam trajectory is compared over several steps.
"""
import numpy as np
import pytest
import torch

from oracle import vae_oracle as O

pytestmark = pytest.mark.gpu


def _engine_mod():
    from conv1d_text_vae_amd import engine, geometry
    return engine, geometry


from parity import assert_activation, rel_err, rel_l2      # noqa: E402


CASES = {
    'tiny1': dict(B=4, T_in=12, T_out=11, E=8, n_filters=(6,), k=3, D=5, N=17),
    'deep3': dict(B=3, T_in=30, T_out=26, E=30, n_filters=(5, 7, 4), k=3, D=9, N=70),
    'k4': dict(B=2, T_in=15, T_out=14, E=5, n_filters=(4, 6), k=4, D=3, N=11),
    'k2': dict(B=5, T_in=9, T_out=9, E=40, n_filters=(33,), k=2, D=3, N=130),
    'mid': dict(B=16, T_in=20, T_out=18, E=62, n_filters=(48, 40), k=3, D=24, N=300),
    'wide': dict(B=8, T_in=16, T_out=16, E=100, n_filters=(32,), k=3, D=16, N=500),   # kp=128 (100-d embeddings)
    'kp192': dict(B=6, T_in=14, T_out=13, E=150, n_filters=(24,), k=3, D=12, N=400),
    'kp256': dict(B=6, T_in=14, T_out=15, E=200, n_filters=(24,), k=3, D=12, N=700),  # 200-d embeddings
    'kp384': dict(B=4, T_in=12, T_out=12, E=340, n_filters=(16,), k=3, D=8, N=300),   # wider than any fused kernel: unfused path
    'big128': dict(B=2048, T_in=32, T_out=32, E=62, n_filters=(128,), k=3, D=32, N=200),   # 128x128 GEMM tiles (>= 384 of them)
    'kp320': dict(B=6, T_in=14, T_out=13, E=316, n_filters=(64,), k=3, D=12, N=300),   # V = 318: the widest fused row (kp = 320), whole-row out conv in bf16
    'ragged': dict(B=37, T_in=11, T_out=9, E=62, n_filters=(64,), k=3, D=8, N=150),    # 37 samples x odd lengths: whole-row tiles straddle samples and end ragged
    'longflat': dict(B=8, T_in=70, T_out=20, E=30, n_filters=(64,), k=3, D=16, N=90),   # flat = 2176: split-K heads GEMM (8 slabs)
    'k4f16': dict(B=5, T_in=15, T_out=14, E=12, n_filters=(16,), k=4, D=12, N=40),     # 4 taps, 5 samples (fused decoder head: 4 per workgroup)
    'k2f24': dict(B=7, T_in=13, T_out=12, E=20, n_filters=(24, 8), k=2, D=70, N=60),   # 2 taps, two layers, latent_dim > 64
}


def make_case(c, seed=0):
    V = c['E'] + 2
    og = O.VAEGeometry(c['T_in'], c['T_out'], V, V, c['n_filters'], c['k'], c['D'])
    params = O.init_params(og, seed=3 + seed, dtype=np.float32)
    rng = np.random.RandomState(seed)
    for name in params:
        if name.endswith('/bias'):
            params[name] = rng.uniform(-0.1, 0.1, params[name].shape).astype(np.float32)
    Wv = O.synthetic_vocab(c['N'] - 2, c['E'], seed=5 + seed)
    x, ids = O.synthetic_batch(Wv, Wv.shape[0], c['B'], c['T_in'], c['T_out'], seed=7 + seed)
    eps = rng.standard_normal((c['B'], c['D'])).astype(np.float32)
    mask_e = (rng.uniform(size=(c['B'], og.flat_size)) < 0.7)
    mask_d = (rng.uniform(size=(c['B'], c['D'])) < 0.7)
    return og, params, Wv, x, ids, eps, mask_e, mask_d


def build_engine(c, og, params, Wv, dtype):
    engine, geometry = _engine_mod()
    V = c['E'] + 2
    g = geometry.Geometry(c['T_in'], c['T_out'], V, V, c['n_filters'], c['k'], c['D'])
    return engine.VAEEngine(g, c['B'], Wv, dtype=dtype, weights=params)


def load_fixed(eng, x, ids, eps, mask_e, mask_d):
    eng.load_batch(x, ids)
    eng.eps.copy_(torch.from_numpy(eps))
    eng.keep_e.copy_(torch.from_numpy(mask_e.astype(np.uint8)))
    eng.keep_d.copy_(torch.from_numpy(mask_d.astype(np.uint8)))


@pytest.mark.parametrize('name', list(CASES))
@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_train_forward_backward_parity(name, dtype):
    c = CASES[name]
    og, params, Wv, x, ids, eps, mask_e, mask_d = make_case(c)
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    loss, grads, acts = O.train_forward_backward(og, p64, Wv.astype(np.float64), x.astype(np.float64), ids,
                                                 eps.astype(np.float64), mask_e, mask_d)
    eng = build_engine(c, og, params, Wv, dtype)
    if name == 'longflat':
        assert eng.heads_splits == 8
    if dtype == 'bf16':
        assert (eng.logits is None) == (name != 'kp384'), 'fused stream kernel expected for every width up to 320'
    load_fixed(eng, x, ids, eps, mask_e, mask_d)
    eng._run(eng.train_ops(False))
    torch.cuda.synchronize()
    tol = 1e-4 if dtype == 'f32' else 1e-2
    D = c['D']
    mlv = eng.mlv.cpu().numpy()
    assert_activation(mlv[:, :D], acts['z_mean'], dtype, 'z_mean')
    assert_activation(mlv[:, D:], acts['z_log_var'], dtype, 'z_log_var')
    assert_activation(eng.z_out.cpu().numpy(), acts['z'], dtype, 'z')
    assert_activation(eng.y_pred().cpu().numpy(), acts['y_pred'], dtype, 'y_pred')
    got_loss = eng.loss()
    assert abs(got_loss - loss) < (1e-4 if dtype == 'f32' else 1e-2) * max(1.0, abs(loss)), (got_loss, loss)
    np.testing.assert_allclose(eng.xent_b.cpu().numpy(), acts['xent_b'], rtol=tol, atol=tol)
    g = eng.get_grads()
    for k_, ref in grads.items():
        assert g[k_].shape == ref.shape
        if dtype == 'f32':
            assert rel_err(g[k_], ref) < 2e-4, (k_, rel_err(g[k_], ref))
        else:
            # bf16 storage can flip a ReLU / max-pool decision of a near-tie, which is a discrete change in a net
            # this small: bound the Frobenius error tightly and the worst element loosely
            assert rel_l2(g[k_], ref) < 0.1, (k_, rel_l2(g[k_], ref))
            assert rel_err(g[k_], ref) < 0.3, (k_, rel_err(g[k_], ref))


def test_stream_kernel_whole_rounds_plus_split_remainder(monkeypatch):
    """More live row tiles than CUs (the C4 regime, here with a toy model: 1400 samples x 64 steps, ~350 live row tiles):
    c1v_live_rows deals the first 256 row tiles one workgroup each over the whole vocabulary and cuts the remainder into
    vocabulary splits (ctl[4], ctl[2]); the finish step sums one slab for the former and ctl[2] for the latter.  Loss,
    per-sample cross-entropy and the decoder-side gradients against the fp64 oracle, and against the same engine without
    the live-row compaction (static grid, every row through the kernel)."""
    c = dict(B=1400, T_in=20, T_out=64, E=30, n_filters=(8,), k=3, D=4, N=320)
    og, params, Wv, x, ids, eps, mask_e, mask_d = make_case(c)
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    loss, grads, acts = O.train_forward_backward(og, p64, Wv.astype(np.float64), x.astype(np.float64), ids,
                                                 eps.astype(np.float64), mask_e, mask_d)
    res = {}
    for live in ('1', '0'):
        monkeypatch.setenv('C1V_LIVE_ROWS', live)
        eng = build_engine(c, og, params, Wv, 'bf16')
        load_fixed(eng, x, ids, eps, mask_e, mask_d)
        eng._run(eng.train_ops(False))
        torch.cuda.synchronize()
        if live == '1':
            ctl = eng.live_ctl.cpu().numpy()
            n_live = int((ids >= 0).sum())
            assert ctl[0] == n_live and ctl[1] == -(-n_live // 128)
            assert ctl[1] > 256, 'the case must have more live row tiles than CUs'
            assert ctl[4] == 256 and ctl[2] >= 2, ctl
            assert ctl[4] + (ctl[1] - ctl[4]) * ctl[2] <= (eng.m_pad // 128) * eng.x_splits
        res[live] = (eng.loss(), eng.xent_b.cpu().numpy(), eng.get_grads())
    for live, (l, xb, g) in res.items():
        assert abs(l - loss) < 1e-2 * max(1.0, abs(loss)), (live, l, loss)
        np.testing.assert_allclose(xb, acts['xent_b'], rtol=1e-2, atol=1e-2)
        for k_ in ('deconv_decoder_embeddings/kernel', 'deconv_decoder_embeddings/bias'):
            assert rel_l2(g[k_], grads[k_]) < 2e-2, (live, k_, rel_l2(g[k_], grads[k_]))
    assert abs(res['1'][0] - res['0'][0]) < 2e-3 * max(1.0, abs(loss))
    np.testing.assert_allclose(res['1'][1], res['0'][1], rtol=5e-3, atol=5e-3)


@pytest.mark.parametrize('name', ['tiny1', 'deep3', 'k4', 'mid', 'kp192', 'big128'])
@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_large_gemm_variant_parity(name, dtype, monkeypatch):
    """The large-GEMM variant of the NT kernel (128 x 128 tile, two-stage ring, two workgroups per CU, epilogue inputs
    requested after the K loop) is normally taken from 256 tiles on -- the C4 stack at batch 2048.  Forced here onto every
    GEMM of the small cases (all epilogue modes, ragged tiles, odd channel counts): the same parity bars against the
    oracle, and bit-identical results to the other variants (64 x 64, and 128 x 128 with a three-stage ring on `big128`;
    same MFMA order per output element), which the second run forces by putting the threshold out of reach."""
    c = CASES[name]
    og, params, Wv, x, ids, eps, mask_e, mask_d = make_case(c)
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    loss, grads, acts = O.train_forward_backward(og, p64, Wv.astype(np.float64), x.astype(np.float64), ids,
                                                 eps.astype(np.float64), mask_e, mask_d)
    got = {}
    for mode in ('1', None):
        monkeypatch.setenv('C1V_NT_LARGE_TILES', mode if mode is not None else '1000000000')
        eng = build_engine(c, og, params, Wv, dtype)
        load_fixed(eng, x, ids, eps, mask_e, mask_d)
        eng._run(eng.train_ops(False))
        torch.cuda.synchronize()
        got[mode] = (eng.loss(), eng.y_pred().cpu().numpy(), eng.mlv.cpu().numpy(), eng.get_grads())
    l, y, mlv, g = got['1']
    assert_activation(y, acts['y_pred'], dtype, 'y_pred')
    assert_activation(mlv[:, :c['D']], acts['z_mean'], dtype, 'z_mean')
    assert abs(l - loss) < (1e-4 if dtype == 'f32' else 1e-2) * max(1.0, abs(loss))
    for k_, ref in grads.items():
        if dtype == 'f32':
            assert rel_err(g[k_], ref) < 2e-4, (k_, rel_err(g[k_], ref))
        else:
            assert rel_l2(g[k_], ref) < 0.1, (k_, rel_l2(g[k_], ref))
    l0, y0, mlv0, g0 = got[None]
    assert l == l0 and np.array_equal(y, y0) and np.array_equal(mlv, mlv0)
    for k_ in g0:
        np.testing.assert_array_equal(g[k_], g0[k_], err_msg=k_)


@pytest.mark.parametrize('name', ['deep3', 'mid'])
@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_paired_backward_launch_is_bit_identical(name, dtype, monkeypatch):
    """c1v_bwd_pair (dgrad + wgrad tiles in one grid) against the two separate entry points: same kernels bodies, same
    summation order, so the gradients must agree to the bit."""
    c = CASES[name]
    og, params, Wv, x, ids, eps, mask_e, mask_d = make_case(c)
    got = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('C1V_BWD_PAIR', mode)
        eng = build_engine(c, og, params, Wv, dtype)
        load_fixed(eng, x, ids, eps, mask_e, mask_d)
        eng._run(eng.train_ops(False))
        torch.cuda.synchronize()
        got[mode] = (eng.get_grads(), eng.loss())
    assert got['1'][1] == got['0'][1]
    for k_, v in got['0'][0].items():
        np.testing.assert_array_equal(got['1'][0][k_], v, err_msg=k_)


@pytest.mark.parametrize('name', ['deep3', 'k4', 'k2', 'mid', 'big128'])
@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_fused_tap_wgrad_is_bit_identical(name, dtype, monkeypatch):
    """c1v_wgrad_args::tap_group (the taps of a convolution's weight gradient in ONE workgroup: both operands fetched once
    for all taps; taken by itself for large batches, forced here) against one workgroup per tap: every output element sums
    the same products in the same order, so the gradients must agree to the bit (k = 2, 3, 4; k = 4 runs two groups of
    two taps)."""
    c = CASES[name]
    og, params, Wv, x, ids, eps, mask_e, mask_d = make_case(c)
    monkeypatch.setenv('C1V_WGRAD_MAX_SPLITS', '8')          # the same row splits both ways
    got = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('C1V_WGRAD_FUSE', mode)
        eng = build_engine(c, og, params, Wv, dtype)
        assert all((tg >= 2) == (mode == '1') for tg in eng.wgrad_fused.values()), eng.wgrad_fused
        load_fixed(eng, x, ids, eps, mask_e, mask_d)
        eng._run(eng.train_ops(False))
        torch.cuda.synchronize()
        got[mode] = (eng.get_grads(), eng.loss())
    assert got['1'][1] == got['0'][1]
    for k_, v in got['0'][0].items():
        np.testing.assert_array_equal(got['1'][0][k_], v, err_msg=k_)


@pytest.mark.parametrize('name', ['deep3', 'k4', 'k2', 'mid', 'wide', 'big128'])
@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_wide_window_nt_variant_is_bit_identical(name, dtype, monkeypatch):
    """The wide NT variant (256 x 128 tiles over the flat buffer rows, the activation window of a channel strip fetched
    once for all taps; taken by itself for the large convolutions of C4, forced here with C1V_NTW=1 onto every convolution
    whose channel pitch is whole 128-byte strips) against the 64 x 64 / 128 x 128 kernels: the same products summed in
    the same order, so activations, loss and gradients must agree to the bit -- all epilogue modes (pool, upsample + crop,
    pair sum, un-pool), k = 2, 3, 4, ragged tiles, samples that straddle tiles."""
    c = CASES[name]
    og, params, Wv, x, ids, eps, mask_e, mask_d = make_case(c)
    got = {}
    for mode in ('1', '2', '0'):                 # 128 x 128 (two workgroups per CU) | 256 x 128 (one) | the plain kernels
        monkeypatch.setenv('C1V_NTW', mode)
        eng = build_engine(c, og, params, Wv, dtype)
        load_fixed(eng, x, ids, eps, mask_e, mask_d)
        eng._run(eng.train_ops(False))
        torch.cuda.synchronize()
        got[mode] = (eng.get_grads(), eng.loss(), eng.y_pred().cpu().numpy(), eng.mlv.cpu().numpy())
    for mode in ('1', '2'):
        assert got[mode][1] == got['0'][1]
        np.testing.assert_array_equal(got[mode][2], got['0'][2])
        np.testing.assert_array_equal(got[mode][3], got['0'][3])
        for k_, v in got['0'][0].items():
            np.testing.assert_array_equal(got[mode][0][k_], v, err_msg=k_)


@pytest.mark.parametrize('name', ['deep3', 'k4', 'k2', 'mid', 'wide', 'big128', 'kp320', 'ragged'])
@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_output_convolution_with_fused_l2norm(name, dtype, monkeypatch):
    """C1V_ROW_NORM (the output convolution's workgroup holds whole rows and l2-normalises them in its epilogue: yhat at the
    compact live-row position, rnorm, fp32 y of the live rows) against STORE + c1v_l2norm_rows.  The fp32 rows of y must
    agree to the bit.  The sum of squares of a row is added up in another order (accumulator registers + DPP instead of a
    pass over the stored row), so 1 / |y| may differ in its last bits: the normalised operand within one rounding step of
    its type, loss and gradients within 1e-6 (f32) / 2e-3 (bf16: single flipped roundings of yhat), and the same for the
    validation loss (eval_step: the same epilogue over ALL rows of y)."""
    c = CASES[name]
    og, params, Wv, x, ids, eps, mask_e, mask_d = make_case(c)
    got = {}
    monkeypatch.setenv('C1V_NORM_LIVE_ONLY', '1')         # the product default (conftest.py turns it off for the other tests)
    for mode in ('1', '0'):
        monkeypatch.setenv('C1V_FUSE_NORM', mode)
        eng = build_engine(c, og, params, Wv, dtype)
        if mode == '1' and not eng.fuse_norm:
            pytest.skip('this shape does not take the whole-row kernel (%d-byte channel rows)' % (eng.Xf.cp * (2 if dtype == 'bf16' else 4)))
        assert eng.fuse_norm == (mode == '1')
        load_fixed(eng, x, ids, eps, mask_e, mask_d)
        eng._run(eng.train_ops(False))
        torch.cuda.synchronize()
        live = (ids.reshape(-1) >= 0) if eng.compact else np.ones(ids.size, bool)
        y = eng.y_pred().cpu().numpy().reshape(ids.size, -1)[live]
        n_rows = int(live.sum()) if eng.compact else ids.size
        yhat = eng.yhat[:n_rows * eng.kp].float().cpu().numpy().reshape(n_rows, eng.kp)
        rn = eng.rnorm[:ids.size].cpu().numpy()[live]
        res = [eng.get_grads(), eng.loss(), y, yhat, rn]
        eng.eval_step(use_graph=False, fixed_rng=True)
        torch.cuda.synchronize()
        res += [eng.loss(), eng.y_pred().cpu().numpy()]
        got[mode] = res
    a, b = got['1'], got['0']
    np.testing.assert_array_equal(a[2], b[2])
    np.testing.assert_array_equal(a[6], b[6])
    np.testing.assert_allclose(a[4], b[4], rtol=1e-6, atol=0)
    step = 2.0 ** -7 if dtype == 'bf16' else 4e-7
    assert np.all(np.abs(a[3] - b[3]) <= step * np.abs(b[3]) + 1e-30)
    assert np.all(a[3][:, eng.V:] == 0)
    tol = 2e-3 if dtype == 'bf16' else 1e-6
    assert abs(a[1] - b[1]) <= tol * abs(b[1]) and abs(a[5] - b[5]) <= tol * abs(b[5])
    for k_, v in b[0].items():
        assert rel_l2(a[0][k_], v) < tol, (k_, rel_l2(a[0][k_], v))


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_step_prologue_and_live_rows_are_bit_identical(dtype, monkeypatch):
    """One-launch step prologue (rng + input pack + live-row table) and the compacted loss rows against the separate
    launches over all rows: same draws (same Philox counters), same arithmetic per live row -> same loss and gradients."""
    c = CASES['mid']
    og, params, Wv, x, ids, eps, mask_e, mask_d = make_case(c)
    assert (ids < 0).any() and (ids >= 0).any()
    got = {}
    for mode in (('1', '1'), ('0', '1'), ('0', '0')):
        monkeypatch.setenv('C1V_PROLOGUE', mode[0])
        monkeypatch.setenv('C1V_LIVE_ROWS', mode[1])
        eng = build_engine(c, og, params, Wv, dtype)
        eng.load_batch(x, ids)
        eng._run(eng.train_ops(True))
        torch.cuda.synchronize()
        got[mode] = (eng.get_grads(), eng.loss(), eng.eps.cpu().numpy().copy(), eng.keep_e.cpu().numpy().copy())
    ref = got[('0', '0')]
    for mode in (('1', '1'), ('0', '1')):
        np.testing.assert_array_equal(got[mode][2], ref[2])
        np.testing.assert_array_equal(got[mode][3], ref[3])
        if mode == ('1', '1'):                       # prologue on/off with the same loss path: to the bit
            assert got[mode][1] == got[('0', '1')][1]
            for k_, v in got[('0', '1')][0].items():
                np.testing.assert_array_equal(got[mode][0][k_], v, err_msg=k_)
        # compacted vs all rows: the vocabulary splits differ, so sums associate differently -- close, not identical
        assert abs(got[mode][1] - ref[1]) < 1e-5 * max(1.0, abs(ref[1]))
        for k_, v in ref[0].items():
            assert rel_err(got[mode][0][k_], v) < (1e-4 if dtype == 'f32' else 2e-2), k_


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_nadam_trajectory_parity(dtype):
    """Five optimizer steps with fixed eps / masks: parameters and loss follow the oracle's Nadam(clipnorm=10)."""
    c = CASES['mid']
    og, params, Wv, x, ids, eps, mask_e, mask_d = make_case(c, seed=1)
    ref_p = {k: v.astype(np.float64) for k, v in params.items()}
    st = O.NadamState()
    eng = build_engine(c, og, params, Wv, dtype)
    load_fixed(eng, x, ids, eps, mask_e, mask_d)
    ref_losses, got_losses = [], []
    for step in range(5):
        loss, grads, _ = O.train_forward_backward(og, ref_p, Wv.astype(np.float64), x.astype(np.float64), ids,
                                                  eps.astype(np.float64), mask_e, mask_d)
        norm = O.nadam_step(ref_p, grads, st)
        ref_losses.append(loss)
        eng.train_step(use_graph=False, fixed_rng=True)
        got_losses.append(eng.loss())
        gn = float(eng.norm_out.item())
        assert abs(gn - norm) < (1e-3 if dtype == 'f32' else 5e-2) * norm
    w = eng.get_weights()
    tol = 1e-4 if dtype == 'f32' else 3e-2
    for k_, ref in ref_p.items():
        # compare the UPDATE (w - w0), which is what the optimizer produced
        upd_ref = ref - params[k_].astype(np.float64)
        upd_got = w[k_].astype(np.float64) - params[k_].astype(np.float64)
        if dtype == 'f32':
            assert rel_err(upd_got, upd_ref) < 5e-3, (k_, rel_err(upd_got, upd_ref))
        else:
            # Nadam divides by sqrt(v): an element whose gradient is at the bf16 noise floor still moves by ~lr, with a
            # noisy sign, so the update is compared in the Frobenius norm
            assert rel_l2(upd_got, upd_ref) < 0.3, (k_, rel_l2(upd_got, upd_ref))
    np.testing.assert_allclose(got_losses, ref_losses, rtol=1e-3 if dtype == 'f32' else 1e-2)


@pytest.mark.parametrize('name', ['deep3', 'k4', 'mid', 'wide', 'big128'])
@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_xcd_aware_deal_of_the_paired_wgrad_workgroups_is_bit_identical(name, dtype, monkeypatch):
    """C1V_TN_XCD: the wgrad workgroups of a paired backward launch dealt so that the tiles and taps of one (row split, bx)
    run on one XCD (padding workgroups return at once) against the plain deal: the same tiles, the same sums."""
    c = CASES[name]
    og, params, Wv, x, ids, eps, mask_e, mask_d = make_case(c)
    got = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('C1V_TN_XCD', mode)
        eng = build_engine(c, og, params, Wv, dtype)
        load_fixed(eng, x, ids, eps, mask_e, mask_d)
        eng._run(eng.train_ops(False))
        torch.cuda.synchronize()
        got[mode] = (eng.get_grads(), eng.loss(), eng.mlv.cpu().numpy())
    assert got['1'][1] == got['0'][1]
    np.testing.assert_array_equal(got['1'][2], got['0'][2])
    for k_, v in got['0'][0].items():
        np.testing.assert_array_equal(got['1'][0][k_], v, err_msg=k_)


@pytest.mark.parametrize('name', ['tiny1', 'deep3', 'k4', 'k2', 'mid', 'wide', 'kp384', 'longflat'])
@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_all_weight_gradients_in_one_launch_are_bit_identical(name, dtype, monkeypatch):
    """c1v_wgrad_batch (small batches: the chain of data gradients as launches of their own, then every weight gradient of
    the model in ONE launch) against one paired dgrad + wgrad launch per layer: the same tiles and the same sums, so loss
    and gradients agree to the bit -- as single ops and as a recorded step (three Nadam steps, parameters equal)."""
    c = CASES[name]
    og, params, Wv, x, ids, eps, mask_e, mask_d = make_case(c)
    got = {}
    monkeypatch.setenv('C1V_WGRAD_ROWS', '384')            # (the same row splits both ways: the batched form defaults to longer ones)
    # batched with the first wgrad / every wgrad but the last forked to the second stream (c1v_run_list_fork), batched on one
    # stream, and the pairs
    for mode in ('1', '1/2', '1/0', '1/f', '0'):
        monkeypatch.setenv('C1V_WGRAD_BATCH', mode[0])
        monkeypatch.setenv('C1V_WGRAD_OVERLAP', '0' if mode == '1/f' else (mode[2:] or '1'))    # ('1': first wgrad forked; the product default is 0)
        monkeypatch.setenv('C1V_WGRAD_BATCH_FUSE', '1' if mode == '1/f' else '0')            # ('1/f': fused-tap problems inside the batch)
        eng = build_engine(c, og, params, Wv, dtype)
        assert (eng.bwd_batch is not None) == (mode[0] == '1')
        load_fixed(eng, x, ids, eps, mask_e, mask_d)
        eng._run(eng.train_ops(False))
        torch.cuda.synchronize()
        grads, loss = eng.get_grads(), eng.loss()
        for _ in range(3):
            eng.train_step(use_graph=True, fixed_rng=True)
        torch.cuda.synchronize()
        got[mode] = (grads, loss, eng.params.cpu().numpy().copy())
    for mode in ('1', '1/2', '1/0', '1/f'):
        assert got[mode][1] == got['0'][1]
        for k_, v in got['0'][0].items():
            np.testing.assert_array_equal(got[mode][0][k_], v, err_msg=k_)
        np.testing.assert_array_equal(got[mode][2], got['0'][2], err_msg=mode)


@pytest.mark.parametrize('name', ['tiny1', 'k4f16', 'k2f24', 'mid', 'wide', 'ragged', 'longflat', 'big128'])
def test_decoder_head_in_one_launch_is_bit_identical(name, monkeypatch):
    """c1v_dec_head_fwd (latent head -> decoder dense -> first transposed convolution in ONE launch) against the three launches
    it replaces, in the recorded step and in the recorded inference / validation passes (no dropout mask there): every buffer
    the three write, the loss, the gradients and a three-step Nadam trajectory agree to the bit.  Batches that are not a
    multiple of the kernel's four samples per workgroup, one and several partial slabs of the heads GEMM, 2 to 4 taps."""
    c = CASES[name]
    og, params, Wv, x, ids, eps, mask_e, mask_d = make_case(c)
    got = {}
    monkeypatch.setenv('C1V_DEC_HEAD_MAX_BATCH', '1000000')         # (big128: several rounds of workgroups; the engine stops at 1024)
    for mode in ('1', '0'):
        monkeypatch.setenv('C1V_DEC_HEAD_FUSE', mode)
        eng = build_engine(c, og, params, Wv, 'bf16')
        load_fixed(eng, x, ids, eps, mask_e, mask_d)
        eng.train_step(use_graph=True, fixed_rng=True)
        torch.cuda.synchronize()
        names = [getattr(o, 'c_name', '') for o in eng._lists['train_fixed'][3]]
        if mode == '1' and 'c1v_dec_head_fwd' not in names:
            pytest.skip('this shape is outside the fused kernel\'s limits')
        assert ('c1v_dec_head_fwd' in names) == (mode == '1') and ('c1v_latent_head_fwd' in names) == (mode == '0')
        bufs = {'Xd0': eng.Xd[0].t, 'Xnext': (eng.Xd[1] if len(eng.Xd) > 1 else eng.Xf).t, 'zd': eng.zd, 'mlv': eng.mlv,
                'z_out': eng.z_out, 'kl_b': eng.kl_b, 'grads': eng.grads, 'loss': eng.loss_out}
        train = {k_: v.float().cpu().numpy().copy() for k_, v in bufs.items()}
        for _ in range(2):
            eng.train_step(use_graph=True, fixed_rng=True)
        eng.eval_step(use_graph=True, fixed_rng=True)
        torch.cuda.synchronize()
        evalb = {k_: bufs[k_].float().cpu().numpy().copy() for k_ in ('Xd0', 'Xnext', 'zd', 'z_out', 'kl_b', 'loss')}
        y = eng.reconstruct(postprocess=False).float().cpu().numpy().copy()
        got[mode] = (train, evalb, y, eng.params.cpu().numpy().copy())
    for part in (0, 1):
        for k_, v in got['0'][part].items():
            np.testing.assert_array_equal(got['1'][part][k_], v, err_msg='%s (%s)' % (k_, ('train', 'eval')[part]))
    np.testing.assert_array_equal(got['1'][2], got['0'][2])
    np.testing.assert_array_equal(got['1'][3], got['0'][3])


@pytest.mark.parametrize('name', ['mid', 'wide', 'big128', 'kp320', 'ragged'])
def test_finish_step_over_the_live_rows_only_is_bit_identical(name, monkeypatch):
    """c1v_xent_finish_args::live_rows (one wave per LIVE row through the inverse live-row table; the zero gradient rows of
    the positions without a target written by the output convolution's epilogue, C1V_ROW_NORM's nrm_dy) against one wave per
    row: the same arithmetic per live row and zeros elsewhere, so the gradient buffer, every gradient, the loss scalars and
    a three-step Nadam trajectory agree to the bit.  The step is run twice with different targets in between, so that rows
    that were live and are not any more must have been zeroed again."""
    c = CASES[name]
    og, params, Wv, x, ids, eps, mask_e, mask_d = make_case(c)
    ids2 = ids.copy()
    ids2[:, ids.shape[1] // 3:] = -1                     # shorter texts: most live rows of `ids` are dead here
    got = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('C1V_FINISH_COMPACT', mode)
        eng = build_engine(c, og, params, Wv, 'bf16')
        if mode == '1' and not eng.finish_compact:
            pytest.skip('this shape does not take the whole-row output convolution')
        assert eng.finish_compact == (mode == '1')
        res = []
        for ids_k in (ids, ids2, ids):
            load_fixed(eng, x, ids_k, eps, mask_e, mask_d)
            eng._run(eng.train_ops(False))
            torch.cuda.synchronize()
            res.append((eng.get_grads(), eng.loss_out.cpu().numpy().copy(), eng.dOf.valid().float().cpu().numpy().copy()))
        for _ in range(3):
            eng.train_step(use_graph=True, fixed_rng=True)
        eng.eval_step(use_graph=False, fixed_rng=True)
        torch.cuda.synchronize()
        got[mode] = (res, eng.params.cpu().numpy().copy(), eng.loss())
    for (ga, la, da), (gb, lb, db) in zip(got['1'][0], got['0'][0]):
        np.testing.assert_array_equal(la, lb)
        np.testing.assert_array_equal(da, db)
        for k_, v in gb.items():
            np.testing.assert_array_equal(ga[k_], v, err_msg=k_)
    np.testing.assert_array_equal(got['1'][1], got['0'][1])
    assert got['1'][2] == got['0'][2]


@pytest.mark.parametrize('name', ['mid', 'deep3'])
def test_loss_scalars_with_the_per_sample_means_in_their_own_launch(name, monkeypatch):
    """Large batches (>= 65 536 target rows) sum the per-step losses of a sample in c1v_loss_sample_means (one wave per
    sample) instead of inside the single loss workgroup of c1v_grad_finalize_loss; forced here on small cases: the
    gradients are the same bits, the loss scalars agree to rounding (another order of the T additions per sample)."""
    c = CASES[name]
    og, params, Wv, x, ids, eps, mask_e, mask_d = make_case(c)
    got = {}
    for rows in ('1', '1000000000'):
        monkeypatch.setenv('C1V_LOSS_PRESUM_MIN_ROWS', rows)
        eng = build_engine(c, og, params, Wv, 'bf16')
        assert len(eng.ops_finalize) == (2 if rows == '1' else 1)
        load_fixed(eng, x, ids, eps, mask_e, mask_d)
        eng._run(eng.train_ops(False))
        torch.cuda.synchronize()
        got[rows] = (eng.get_grads(), eng.loss_out.cpu().numpy().copy(), eng.xent_b.cpu().numpy().copy())
    a, b = got['1'], got['1000000000']
    np.testing.assert_allclose(a[1], b[1], rtol=1e-6)
    np.testing.assert_allclose(a[2], b[2], rtol=1e-6)
    for k_, v in b[0].items():
        np.testing.assert_array_equal(a[0][k_], v, err_msg=k_)


@pytest.mark.parametrize('launch', ['list', 'graph'])
def test_graph_replay_matches_eager_and_is_deterministic(launch, monkeypatch):
    """The captured step -- issued from native code by c1v_run_list (the default) or replayed as a HIP graph
    (C1V_STEP_LAUNCH=graph) -- against the same launches issued one by one from Python: same bits, run after run."""
    monkeypatch.setenv('C1V_STEP_LAUNCH', launch)
    c = CASES['mid']
    og, params, Wv, x, ids, eps, mask_e, mask_d = make_case(c, seed=2)
    res = []
    for use_graph in (False, True, True):
        eng = build_engine(c, og, params, Wv, 'bf16')
        load_fixed(eng, x, ids, eps, mask_e, mask_d)
        for _ in range(3):
            eng.train_step(use_graph=use_graph, fixed_rng=True)
        torch.cuda.synchronize()
        assert not use_graph or (len(eng._lists) > 0 and all(eng._lists.values())) == (launch == 'list'), getattr(eng, '_list_blockers', None)
        assert not use_graph or (len(eng._graphs) > 0) == (launch == 'graph')
        res.append((eng.params.cpu().numpy().copy(), eng.loss()))
    assert np.array_equal(res[0][0], res[1][0]) and res[0][1] == res[1][1]
    assert np.array_equal(res[1][0], res[2][0])


def test_device_rng_step_changes_and_counter_advances():
    c = CASES['mid']
    og, params, Wv, x, ids, eps, mask_e, mask_d = make_case(c, seed=2)
    eng = build_engine(c, og, params, Wv, 'bf16')
    eng.load_batch(x, ids)
    eng.train_step(use_graph=True)
    e1 = eng.eps.cpu().numpy().copy()
    k1 = eng.keep_e.cpu().numpy().copy()
    eng.train_step(use_graph=True)
    e2 = eng.eps.cpu().numpy().copy()
    assert int(eng.step_ctr.item()) == 2
    assert not np.array_equal(e1, e2)
    # the device streams are the oracle's Philox streams (keyed by global element index)
    ref_eps = O.philox_normal(c['B'] * c['D'], seed=eng.seed, stream=1, offset=0).reshape(c['B'], c['D'])
    np.testing.assert_allclose(e1, ref_eps, rtol=2e-4, atol=2e-5)
    ref_keep = O.philox_bernoulli(c['B'] * og.flat_size, 0.7, seed=eng.seed, stream=2, offset=0).reshape(c['B'], -1)
    assert np.array_equal(k1, ref_keep)
    assert np.isfinite(eng.loss())


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('case', ['mid', 'longflat'])
def test_inference_encode_reconstruct_postprocess(dtype, case):
    c = CASES[case]
    og, params, Wv, x, ids, eps, mask_e, mask_d = make_case(c, seed=4)
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    zm, zl = O.encoder_forward(og, p64, x.astype(np.float64))
    y = O.decoder_forward(og, p64, zm)
    eng = build_engine(c, og, params, Wv, dtype)
    eng.load_batch(x, ids)
    tol = 1e-4 if dtype == 'f32' else 1e-2
    z = eng.encode().cpu().numpy()
    assert_activation(z, zm, dtype, 'transform')
    post = eng.reconstruct(postprocess=True).cpu().numpy()
    ygpu = eng.y_pred().cpu().numpy()
    assert_activation(ygpu, y, dtype, 'reconstruct')
    # post-processing is checked against the oracle applied to the GPU's own y (argmax is discontinuous)
    ref_post = O.postprocess_reconstructed_word_vectors(ygpu)
    np.testing.assert_allclose(post, ref_post, rtol=1e-5, atol=1e-6)
    assert np.array_equal(eng.end_idx.cpu().numpy(), O.eos_end_index(ygpu))
    # vae_decoder_.predict_on_batch(z) alone (the split-K slabs of the heads GEMM must not leak into it)
    y2 = eng.decode_latent(torch.from_numpy(zm.astype(np.float32)).to(eng.dev)).cpu().numpy()
    assert rel_err(y2, y) < tol


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_eval_loss_parity(dtype):
    c = CASES['deep3']
    og, params, Wv, x, ids, eps, mask_e, mask_d = make_case(c, seed=5)
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    ref = O.eval_loss(og, p64, Wv.astype(np.float64), x.astype(np.float64), ids, eps.astype(np.float64))
    eng = build_engine(c, og, params, Wv, dtype)
    load_fixed(eng, x, ids, eps, mask_e, mask_d)
    eng.eval_step(use_graph=False, fixed_rng=True)
    assert abs(eng.loss() - ref) < (1e-4 if dtype == 'f32' else 1e-2) * max(1.0, abs(ref))


def test_eval_loss_all_rows_with_256_row_workgroups(monkeypatch):
    """Validation loss over ALL rows (live-row compaction off) of a batch large enough for the 64-rows-per-wave variant of
    the GEMM1-only stream kernel (SUMEXP mode; the arg-max mode of that variant is pinned by the C3 self-retrieval test)."""
    monkeypatch.setenv('C1V_LIVE_ROWS', '0')
    c = CASES['big128']
    og, params, Wv, x, ids, eps, mask_e, mask_d = make_case(c, seed=6)
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    ref = O.eval_loss(og, p64, Wv.astype(np.float64), x.astype(np.float64), ids, eps.astype(np.float64))
    eng = build_engine(c, og, params, Wv, 'bf16')
    assert not eng.compact and (eng.m_pad // 256) * eng.x_splits >= 256
    load_fixed(eng, x, ids, eps, mask_e, mask_d)
    eng.eval_step(use_graph=False, fixed_rng=True)
    assert abs(eng.loss() - ref) < 1e-2 * max(1.0, abs(ref))


def test_decode_latent_right_after_a_train_step_uses_the_updated_weights():
    """The graph-replayed train step leaves the refresh of the weights' operand copies to the NEXT step's prologue;
    vae_decoder_.predict_on_batch called in between must not run on copies that are one optimizer step old."""
    c = CASES['mid']
    og, params, Wv, x, ids, eps, mask_e, mask_d = make_case(c, seed=8)
    eng = build_engine(c, og, params, Wv, 'f32')
    load_fixed(eng, x, ids, eps, mask_e, mask_d)
    for _ in range(2):
        eng.train_step(use_graph=True, fixed_rng=True)
    z = np.random.RandomState(1).standard_normal((c['B'], c['D'])).astype(np.float32)
    y = eng.decode_latent(torch.from_numpy(z).to(eng.dev)).cpu().numpy()
    w_now = {k: v.astype(np.float64) for k, v in eng.get_weights().items()}
    ref = O.decoder_forward(og, w_now, z.astype(np.float64))
    assert rel_err(y, ref) < 1e-4
    old = O.decoder_forward(og, {k: v.astype(np.float64) for k, v in params.items()}, z.astype(np.float64))
    assert rel_err(old, ref) > 1e-3                    # the step did move the decoder: stale copies would have shown


def test_frozen_encoder_keeps_encoder_weights():
    engine, geometry = _engine_mod()
    c = CASES['mid']
    og, params, Wv, x, ids, eps, mask_e, mask_d = make_case(c, seed=6)
    V = c['E'] + 2
    g = geometry.Geometry(c['T_in'], c['T_out'], V, V, c['n_filters'], c['k'], c['D'])
    eng = engine.VAEEngine(g, c['B'], Wv, dtype='f32', weights=params, frozen_encoder=True)
    load_fixed(eng, x, ids, eps, mask_e, mask_d)
    eng.train_step(use_graph=False, fixed_rng=True)
    w = eng.get_weights()
    for k_ in params:
        same = np.array_equal(w[k_], params[k_])
        if k_.startswith('encoder_conv1d_') or k_.startswith('z_mean') or k_.startswith('z_log_var'):
            assert same, k_
        else:
            assert not same, k_
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    _, grads, _ = O.train_forward_backward(og, p64, Wv.astype(np.float64), x.astype(np.float64), ids,
                                           eps.astype(np.float64), mask_e, mask_d, frozen_encoder=True)
    gg = eng.get_grads()
    for k_, ref in grads.items():
        assert rel_err(gg[k_], ref) < 2e-4 or np.abs(ref).max() == 0 and np.abs(gg[k_]).max() == 0, k_


def test_two_rank_bucketed_step_equals_single_rank_step():
    """Data parallel by construction: two engines (rank 0 / 1 of world 2, half the batch each, on one GPU) stepped through
    the bucketed sequence of `train_step` (A1 | all-reduce decoder bucket | A2 | all-reduce encoder bucket | B) end up
    with the weights of ONE engine stepping the whole batch (fp32; the only difference is the summation order)."""
    engine, geometry = _engine_mod()
    c = dict(CASES['mid'])
    og, params, Wv, x, ids, eps, mask_e, mask_d = make_case(c, seed=9)
    V = c['E'] + 2
    g = geometry.Geometry(c['T_in'], c['T_out'], V, V, c['n_filters'], c['k'], c['D'])
    full = engine.VAEEngine(g, c['B'], Wv, dtype='f32', weights=params)
    load_fixed(full, x, ids, eps, mask_e, mask_d)
    h = c['B'] // 2
    ranks = []
    for r in range(2):
        e = engine.VAEEngine(g, h, Wv, dtype='f32', weights=params, world_size=2, rank=r)
        sl = slice(r * h, (r + 1) * h)
        load_fixed(e, x[sl], ids[sl], eps[sl], mask_e[sl], mask_d[sl])
        ranks.append(e)
    bk = ranks[0].dp_buckets
    assert len(bk) == 3 and bk[0][2][1] == ranks[0].P and bk[-1][2][0] == 0
    assert all(bk[i][2][0] == bk[i + 1][2][1] for i in range(len(bk) - 1))          # contiguous slices, reverse layer order
    for step in range(3):
        full.train_step(use_graph=False, fixed_rng=True)
        # the two "processes" interleaved at their collectives; the all-reduce is a sum of the two ranks' slices
        for i, (hi_stage, _, (lo, hi)) in enumerate(bk):
            lo_stage = bk[i - 1][0] if i else 0
            for e in ranks:
                e._run((e.forward_ops(False, advance=True) if i == 0 else []) +
                       [lambda e=e: e._run_backward(lo_stage, hi_stage), e.dp_buckets[i][1]])
            tot = ranks[0].grads[lo:hi] + ranks[1].grads[lo:hi]
            for e in ranks:
                e.grads[lo:hi].copy_(tot)
        for e in ranks:
            e._run(e.update_ops())
        torch.cuda.synchronize()
        np.testing.assert_allclose(ranks[0].grads.cpu().numpy(), full.grads.cpu().numpy(), rtol=2e-4, atol=1e-7)
    w_full, w0, w1 = full.get_weights(), ranks[0].get_weights(), ranks[1].get_weights()
    for k_ in w_full:
        assert np.array_equal(w0[k_], w1[k_]), k_                       # replicas stay bit-identical
        upd = w_full[k_].astype(np.float64) - params[k_]
        assert rel_err(w0[k_].astype(np.float64) - params[k_], upd) < 5e-3, k_
    # and the public entry point runs the same sequence (world 2 emulated with an identity "all-reduce" x2)
    e = engine.VAEEngine(g, h, Wv, dtype='bf16', weights=params, world_size=2, rank=0)
    load_fixed(e, x[:h], ids[:h], eps[:h], mask_e[:h], mask_d[:h])
    calls = []

    def fake_all_reduce(t):
        calls.append(t.numel())
        t.mul_(2.0)

    for _ in range(2):
        e.train_step(use_graph=True, fixed_rng=True, all_reduce=fake_all_reduce)
    torch.cuda.synchronize()
    assert calls[:3] == [hi - lo for _, _, (lo, hi) in e.dp_buckets] and np.isfinite(e.loss())


def test_two_rank_fit_epochs_equal_the_single_rank_fit():
    """The data-parallel fit loop of the estimator (training.run_epochs_dp, as Conv1dTextVAE._fit_vae drives it) with two
    ranks emulated on one GPU: two half-batch engines (rank 0 / 1 of world 2) stepped through the bucketed sequence with
    the all-reduce done by hand, device RNG on (draws are keyed by the GLOBAL sample index), against ONE engine running
    whole batches through the same loop.  Two epochs of three shuffled batches + one evaluation batch: the same epoch
    losses and the same weights (fp32; only the summation order of the gradient differs)."""
    from conv1d_text_vae_amd import synthetic
    from conv1d_text_vae_amd.dist import DataParallel
    from conv1d_text_vae_amd.training import run_epochs_dp
    engine, geometry = _engine_mod()
    c = dict(CASES['mid'])
    B, n_tr, n_ev = c['B'], 3, 1
    og, params, Wv, _, _, _, _, _ = make_case(c, seed=4)
    V = c['E'] + 2
    g = geometry.Geometry(c['T_in'], c['T_out'], V, V, c['n_filters'], c['k'], c['D'])
    _, ids, in_ids = synthetic.synthetic_batch(Wv, Wv.shape[0], B * (n_tr + n_ev), c['T_in'], c['T_out'], seed=21, with_input_ids=True)
    tr_in, tr_out = torch.from_numpy(in_ids[:B * n_tr]).cuda(), torch.from_numpy(ids[:B * n_tr]).cuda()
    ev_in, ev_out = torch.from_numpy(in_ids[B * n_tr:]).cuda(), torch.from_numpy(ids[B * n_tr:]).cuda()

    def fit(engines, dp):
        """engines: the engine of every emulated rank (one entry = the plain fit)."""
        for e in engines:
            e.set_input_table(Wv)
        acc = torch.zeros(4, dtype=torch.float64, device='cuda')
        n = len(engines)

        def load(e, src_in, src_out, first, count):
            e.in_ids.copy_(src_in[first:first + count])
            e.ids_in.copy_(src_out[first:first + count])

        def train_slice(first, count):
            for r, e in enumerate(engines):
                load(e, tr_in, tr_out, first + r * count, count)
            if n == 1:
                engines[0].train_step(use_graph=False, from_ids=True)
            else:
                bk = engines[0].dp_buckets
                for i, (hi_stage, _, (lo, hi)) in enumerate(bk):
                    lo_stage = bk[i - 1][0] if i else 0
                    for e in engines:
                        e._run((e.forward_ops(True, True, advance=True) if i == 0 else []) +
                               [lambda e=e: e._run_backward(lo_stage, hi_stage), e.dp_buckets[i][1]])
                    tot = sum(e.grads[lo:hi] for e in engines)
                    for e in engines:
                        e.grads[lo:hi].copy_(tot)
                for e in engines:
                    e._run(e.update_ops())
                    e._wop_stale = e._fused_prologue()
            acc[0] += sum(e.loss_out[0] for e in engines)
            acc[1] += 1

        def eval_slice(first, count):
            for r, e in enumerate(engines):
                load(e, ev_in, ev_out, first + r * count, count)
                e.eval_step(use_graph=False, from_ids=True)
            acc[2] += sum(e.loss_out[0] for e in engines)
            acc[3] += 1

        def local_sums():
            a = acc.cpu().numpy().copy()
            acc.zero_()
            return a
        e0 = engines[0]
        snap = lambda: [e.params.clone() for e in engines]

        def restore(w):
            for e, v in zip(engines, w):
                e.params.copy_(v)
                e._pack_all()
        hist = run_epochs_dp(dp, B, n_tr, n_ev, 2, train_slice, eval_slice, local_sums, e0.get_lr, e0.set_lr, snap, restore, seed=77)
        torch.cuda.synchronize()
        return hist, e0.get_weights()

    class TwoRanksHere(DataParallel):
        """rank 0's view; the callbacks above carry rank 1 along and hand over the SUM of both ranks' epoch sums"""
        def __init__(self):
            super().__init__(2, 0, 0)

        def sum_over_ranks(self, values):
            a = np.asarray(values, dtype=np.float64).copy()
            a[1] *= 2.0                                   # (batch counts: every rank counts its own)
            a[3] *= 2.0
            return a

        def broadcast_int(self, value):
            return int(value)

    h1, w1 = fit([engine.VAEEngine(g, B, Wv, dtype='f32', weights=params, seed=5)], DataParallel())
    h2, w2 = fit([engine.VAEEngine(g, B // 2, Wv, dtype='f32', weights=params, seed=5, world_size=2, rank=r) for r in range(2)],
                 TwoRanksHere())
    assert len(h1) == len(h2) == 2
    for a, b in zip(h1, h2):
        assert abs(a['loss'] - b['loss']) < 1e-4 * max(1.0, abs(a['loss'])), (a, b)
        assert abs(a['val_loss'] - b['val_loss']) < 1e-4 * max(1.0, abs(a['val_loss'])), (a, b)
        assert a['lr'] == b['lr']
    for k_ in w1:
        upd = w1[k_].astype(np.float64) - params[k_]
        assert rel_err(w2[k_].astype(np.float64) - params[k_], upd) < 5e-3, k_
