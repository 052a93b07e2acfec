"""Shape walk of the reference graph builder (`__create_model`, conv1d_text_vae.py:1224-1281) and the flat parameter
layout (Keras layer names / weight layouts of the reference, so weights stay interchangeable)."""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import numpy as np


BN_WEIGHTS = ('gamma', 'beta', 'moving_mean', 'moving_variance')      # Keras BatchNormalization weight names
BN_MOVING = ('/moving_mean', '/moving_variance')
BN_MOMENTUM, BN_EPS = 0.99, 1e-3                                       # Keras 2.2 defaults


class Geometry:
    def __init__(self, input_text_size: int, output_text_size: int, input_vector_size: int, output_vector_size: int,
                 n_filters, kernel_size: int, latent_dim: int, use_batch_norm: bool = False):
        if isinstance(n_filters, int):
            n_filters = (n_filters,)
        self.input_text_size = int(input_text_size)
        self.output_text_size = int(output_text_size)
        self.input_vector_size = int(input_vector_size)
        self.output_vector_size = int(output_vector_size)
        self.n_filters = tuple(int(f) for f in n_filters)
        self.kernel_size = k = int(kernel_size)
        self.latent_dim = int(latent_dim)
        self.use_batch_norm = bool(use_batch_norm)       # BatchNormalization behind every conv / transposed conv (:1229, :1259)
        # encoder: Conv1D('valid') -> MaxPool1D(2) per layer (:1227-1241)
        self.enc_in_len: List[int] = []
        self.enc_len_conv: List[int] = []
        self.enc_len_pool: List[int] = []
        L = self.input_text_size
        for _ in self.n_filters:
            self.enc_in_len.append(L)
            L = L - k + 1
            if L < 2:
                raise ValueError('The input text size {0} is too small for {1} convolution layer(s) with kernel size '
                                 '{2}.'.format(self.input_text_size, len(self.n_filters), k))
            self.enc_len_conv.append(L)
            L //= 2
            self.enc_len_pool.append(L)
        self.flat_size = L * self.n_filters[-1]                           # Flatten (:1242)
        # decoder (:1247-1281)
        n_t = int(math.ceil((self.output_text_size - k + 1) / 2.0))
        for _ in range(len(self.n_filters) - 1):
            n_t = int(math.ceil((n_t - k + 1) / 2.0))
        if n_t < 1:
            raise ValueError('The output text size {0} is too small for {1} deconvolution layer(s) with kernel size '
                             '{2}.'.format(self.output_text_size, len(self.n_filters), k))
        self.n_times = n_t
        self.width = max(10, int(math.ceil(self.latent_dim / float(n_t))))
        self.dec_filters = [self.n_filters[-1]] + [self.n_filters[i] for i in reversed(range(len(self.n_filters) - 1))]
        self.dec_in_len: List[int] = []
        self.dec_len_deconv: List[int] = []
        M = n_t
        for _ in self.dec_filters:
            self.dec_in_len.append(M)
            M = M + k - 1
            self.dec_len_deconv.append(M)
            M *= 2
        self.dec_full_len = M
        self.pad_left = (k - 1) // 2                                      # Conv1D 'same' (:1279)

    # ---- parameters ---------------------------------------------------------------------------------------
    def keras_shapes(self) -> Dict[str, Tuple[int, ...]]:
        k = self.kernel_size
        s: Dict[str, Tuple[int, ...]] = {}
        c = self.input_vector_size
        for i, f in enumerate(self.n_filters):
            s['encoder_conv1d_%d/kernel' % (i + 1)] = (k, c, f)
            s['encoder_conv1d_%d/bias' % (i + 1)] = (f,)
            if self.use_batch_norm:
                for w in BN_WEIGHTS:
                    s['encoder_batchnorm_%d/%s' % (i + 1, w)] = (f,)
            c = f
        s['z_mean/kernel'] = (self.flat_size, self.latent_dim)
        s['z_mean/bias'] = (self.latent_dim,)
        s['z_log_var/kernel'] = (self.flat_size, self.latent_dim)
        s['z_log_var/bias'] = (self.latent_dim,)
        s['deconv_decoder_dense/kernel'] = (self.latent_dim, self.n_times * self.width)
        s['deconv_decoder_dense/bias'] = (self.n_times * self.width,)
        c = self.width
        for i, f in enumerate(self.dec_filters):
            s['deconv_decoder_%d_deconv1d_part2/kernel' % (i + 1)] = (k, 1, f, c)
            s['deconv_decoder_%d_deconv1d_part2/bias' % (i + 1)] = (f,)
            if self.use_batch_norm:
                for w in BN_WEIGHTS:
                    s['deconv_decoder_batchnorm_%d/%s' % (i + 1, w)] = (f,)
            c = f
        s['deconv_decoder_embeddings/kernel'] = (k, c, self.output_vector_size)
        s['deconv_decoder_embeddings/bias'] = (self.output_vector_size,)
        return s

    def segments(self) -> List[Tuple[str, Tuple[int, ...]]]:
        """Flat-buffer segments.  The two latent heads are stored as ONE [flat, 2*latent] kernel (z_mean | z_log_var)
        so that both heads are a single GEMM; `split_heads` / `merge_heads` convert to the Keras pair."""
        out: List[Tuple[str, Tuple[int, ...]]] = []
        for name, shape in self.keras_shapes().items():
            if name.startswith('z_log_var') or name.endswith(BN_MOVING):       # moving averages are not trainable
                continue
            if name == 'z_mean/kernel':
                out.append(('z_heads/kernel', (self.flat_size, 2 * self.latent_dim)))
            elif name == 'z_mean/bias':
                out.append(('z_heads/bias', (2 * self.latent_dim,)))
            else:
                out.append((name, shape))
        return out

    def n_params(self) -> int:
        return int(sum(int(np.prod(s)) for _, s in self.segments()))

    def encoder_names(self) -> List[str]:
        return [n for n, _ in self.segments() if n.startswith(('encoder_conv1d_', 'encoder_batchnorm_', 'z_heads'))]

    def bn_layers(self) -> List[Tuple[str, int]]:
        """(Keras layer name, channels) of every BatchNormalization layer, encoder first."""
        if not self.use_batch_norm:
            return []
        return [('encoder_batchnorm_%d' % (i + 1), f) for i, f in enumerate(self.n_filters)] + \
               [('deconv_decoder_batchnorm_%d' % (i + 1), f) for i, f in enumerate(self.dec_filters)]


def glorot_limit(shape: Tuple[int, ...]) -> float:
    """Keras `glorot_uniform` limit (fans as keras.initializers._compute_fans, channels_last)."""
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:
        receptive = int(np.prod(shape[:-2]))
        fan_in = shape[-2] * receptive
        fan_out = shape[-1] * receptive
    return math.sqrt(6.0 / (fan_in + fan_out))


def init_keras_weights(geom: Geometry, seed: int = 0) -> Dict[str, np.ndarray]:
    """Keras defaults: kernels glorot_uniform, biases zero."""
    rng = np.random.Generator(np.random.PCG64(seed))
    w: Dict[str, np.ndarray] = {}
    for name, shape in geom.keras_shapes().items():
        if name.endswith(('/bias', '/beta', '/moving_mean')):
            w[name] = np.zeros(shape, dtype=np.float32)
        elif name.endswith(('/gamma', '/moving_variance')):
            w[name] = np.ones(shape, dtype=np.float32)
        else:
            lim = glorot_limit(shape)
            w[name] = rng.uniform(-lim, lim, size=shape).astype(np.float32)
    return w


def merge_heads(w: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    out = {k: v for k, v in w.items() if not (k.startswith('z_mean') or k.startswith('z_log_var'))}
    out['z_heads/kernel'] = np.concatenate([w['z_mean/kernel'], w['z_log_var/kernel']], axis=1)
    out['z_heads/bias'] = np.concatenate([w['z_mean/bias'], w['z_log_var/bias']], axis=0)
    return out


def split_heads(w: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    out = {k: v for k, v in w.items() if not k.startswith('z_heads')}
    D = w['z_heads/bias'].shape[0] // 2
    out['z_mean/kernel'] = np.ascontiguousarray(w['z_heads/kernel'][:, :D])
    out['z_log_var/kernel'] = np.ascontiguousarray(w['z_heads/kernel'][:, D:])
    out['z_mean/bias'] = np.ascontiguousarray(w['z_heads/bias'][:D])
    out['z_log_var/bias'] = np.ascontiguousarray(w['z_heads/bias'][D:])
    return out
