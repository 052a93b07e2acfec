"""Shared parity helpers of the GPU tests (error measures, the north-star activation bar, engine read-back)."""
import numpy as np

ACT_TOL = {'f32': 1e-4, 'bf16': 1e-2}        # north_star: activations / latents within 1e-4 relative in fp32, 1e-2 in bf16


def rel_err(a, b):
    """Largest element error relative to the tensor's scale (max |a - b| / max |b|)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def row_rel_l2(a, b):
    """Worst sample row: max_r ||a_r - b_r|| / ||b_r|| over the rows (last axis) of the tensor.  A row whose reference norm
    is below 1 % of the root-mean-square row norm is measured against that floor (a near-zero row has no relative scale)."""
    a = np.asarray(a, dtype=np.float64).reshape(-1, np.shape(a)[-1])
    b = np.asarray(b, dtype=np.float64).reshape(-1, np.shape(b)[-1])
    nb = np.linalg.norm(b, axis=1)
    floor = 0.01 * np.sqrt(np.mean(nb * nb))
    return float((np.linalg.norm(a - b, axis=1) / np.maximum(nb, max(floor, 1e-30))).max())


def assert_activation(got, ref, dtype, what=''):
    """The north-star bar for forward activations and latents, per sample row and for the tensor as a whole.  Rows of the
    toy geometries with fewer than 32 elements carry too few terms for a per-row norm to average three bf16 roundings
    (2^-8 each) below 1e-2; they are held to the bar through the L2 error of the whole tensor instead (every model-sized
    case -- C2, C4, `mid`, `wide`, ... -- is held per row)."""
    tol = ACT_TOL[dtype]
    r = row_rel_l2(got, ref) if np.shape(ref)[-1] >= 32 else rel_l2(got, ref)
    e = rel_err(got, ref)
    assert r < tol, '%s: worst row relative L2 error %.3e >= %.0e (%s)' % (what, r, tol, dtype)
    assert e < tol, '%s: max error relative to the tensor scale %.3e >= %.0e (%s)' % (what, e, tol, dtype)
    return r, e


def engine_decisions(eng):
    """ReLU / max-pool choices the engine's backward pass of the last step acted on, in the layout of the oracle's
    `decisions` argument: read back from the pool-selection bytes and from the stored (post-ReLU) activations."""
    g = eng.g
    nl, nd = len(g.n_filters), len(g.dec_filters)
    B = eng.B
    d = {}
    for l in range(nl):
        d['enc_sel_%d' % l] = eng.sel[l].cpu().numpy().astype(bool)
        if l == nl - 1:
            pooled = eng.H.t[:B * eng.flat_p].view(B, eng.flat_p)[:, :eng.flat].reshape(B, g.enc_len_pool[l], g.n_filters[l])
        else:
            pooled = eng.Xe[l + 1].valid()
        d['enc_on_%d' % l] = (pooled.float() > 0).cpu().numpy()
    d['dense_on'] = (eng.Xd[0].valid().float() > 0).cpu().numpy().reshape(B, -1)
    for i in range(nd):
        nxt = eng.Xf if i == nd - 1 else eng.Xd[i + 1]
        on = (nxt.valid().float()[:, ::2] > 0).cpu().numpy()
        full = np.zeros((B, g.dec_len_deconv[i], g.dec_filters[i]), dtype=bool)      # rows cut off by the crop: no gradient
        full[:, :on.shape[1]] = on[:, :g.dec_len_deconv[i]]
        d['dec_on_%d' % i] = full
    return d
