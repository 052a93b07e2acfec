#!/usr/bin/env python
"""Generates the golden vectors tests/golden/vae_*.npz.

The reference (Keras/TensorFlow + gensim) cannot be imported or run in the build image and its own tests hold no
floating-point golden value for the VAE rows a1-a9, so these vectors come from the float64 run of the
CPU restatement in oracle/vae_oracle.py, AFTER that restatement has been cross-checked against an independent
torch-autograd restatement (tests/test_oracle.py).  They freeze the arithmetic: a later change to the oracle or to
the HIP kernels that moves any number shows up against these files ("parity unpinned" against Keras itself, as stated
in DESIGN.md).  Everything is stored in float32 except the scalar losses.

    python tests/golden/make_golden.py        # rewrites the .npz files next to this script
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import vae_oracle as O  # noqa: E402

CASES = {
    'tiny': dict(B=4, T_in=12, T_out=11, E=8, n_filters=(6,), k=3, D=5, N=17, seed=11),
    'deep': dict(B=3, T_in=30, T_out=26, E=30, n_filters=(5, 7, 4), k=3, D=9, N=70, seed=12),
    'even_k': dict(B=2, T_in=15, T_out=14, E=5, n_filters=(4, 6), k=4, D=3, N=11, seed=13),
}
N_STEPS = 3


def build(c):
    V = c['E'] + 2
    og = O.VAEGeometry(c['T_in'], c['T_out'], V, V, c['n_filters'], c['k'], c['D'])
    params = O.init_params(og, seed=c['seed'], dtype=np.float32)
    rng = np.random.RandomState(c['seed'])
    for name in params:
        if name.endswith('/bias'):
            params[name] = rng.uniform(-0.1, 0.1, params[name].shape).astype(np.float32)
    W = O.synthetic_vocab(c['N'] - 2, c['E'], seed=c['seed'] + 1)
    x, ids = O.synthetic_batch(W, W.shape[0], c['B'], c['T_in'], c['T_out'], seed=c['seed'] + 2)
    eps = rng.standard_normal((c['B'], c['D'])).astype(np.float32)
    mask_e = (rng.uniform(size=(c['B'], og.flat_size)) < 0.7)
    mask_d = (rng.uniform(size=(c['B'], c['D'])) < 0.7)
    return og, params, W, x, ids, eps, mask_e, mask_d


def golden(c):
    og, params, W, x, ids, eps, mask_e, mask_d = build(c)
    out = {'in/W': W, 'in/x': x, 'in/ids': ids, 'in/eps': eps, 'in/mask_e': mask_e.astype(np.uint8),
           'in/mask_d': mask_d.astype(np.uint8)}
    for k, v in params.items():
        out['param/' + k] = v
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    W64, x64, eps64 = W.astype(np.float64), x.astype(np.float64), eps.astype(np.float64)
    loss, grads, acts = O.train_forward_backward(og, p64, W64, x64, ids, eps64, mask_e, mask_d)
    out['out/loss'] = np.float64(loss)
    for k in ('z_mean', 'z_log_var', 'z', 'y_pred', 'xent_b'):
        out['out/' + k] = np.asarray(acts[k], dtype=np.float32)
    for k, v in grads.items():
        out['grad/' + k] = v.astype(np.float32)
    out['out/eval_loss'] = np.float64(O.eval_loss(og, p64, W64, x64, ids, eps64))
    zm, _ = O.encoder_forward(og, p64, x64)
    y_inf = O.decoder_forward(og, p64, zm)
    out['out/transform'] = zm.astype(np.float32)
    out['out/reconstruct'] = y_inf.astype(np.float32)
    out['out/postprocessed'] = O.postprocess_reconstructed_word_vectors(y_inf.astype(np.float32))
    st = O.NadamState()
    losses, norms = [], []
    for _ in range(N_STEPS):
        l, g, _ = O.train_forward_backward(og, p64, W64, x64, ids, eps64, mask_e, mask_d)
        norms.append(O.nadam_step(p64, g, st))
        losses.append(l)
    out['nadam/losses'] = np.asarray(losses, dtype=np.float64)
    out['nadam/norms'] = np.asarray(norms, dtype=np.float64)
    for k, v in p64.items():
        out['nadam/' + k] = v.astype(np.float32)
    return out


def main():
    for name, c in CASES.items():
        path = os.path.join(HERE, 'vae_%s.npz' % name)
        np.savez_compressed(path, **{k.replace('/', '__'): v for k, v in golden(c).items()})
        print(path, os.path.getsize(path), 'bytes')
    # Philox4x32-10 known-answer vectors (Random123 kat_vectors: counter/key all zero, and all ones bits) travel as text
    # in tests/test_golden.py; here: the streams the engine draws (eps and keep masks of the first step)
    path = os.path.join(HERE, 'philox_streams.npz')
    np.savez_compressed(path, normal=O.philox_normal(64, seed=1234, stream=1, offset=0),
                        normal_off=O.philox_normal(16, seed=1234, stream=1, offset=37),
                        bernoulli=O.philox_bernoulli(256, 0.7, seed=1234, stream=2, offset=0))
    print(path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
