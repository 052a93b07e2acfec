"""CPU restatement (NumPy) of the Conv1D text-VAE hot path of nsu-ai-team/conv1d-text-vae.

THIS FILE IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and the ``cpu_baseline``
leg of bench.py may import it; the product path (conv1d_text_vae_amd) never does.

PARITY UNPINNED for rows a1-a9: the arithmetic of these rows lives in Keras >= 2.2.0 / TensorFlow 1.x
(reference requirements.txt:4) which is absent from the reference tree and from this image, and no
reference test holds a floating-point golden value for them.  The restatement
below follows the reference call sites line by line and the published Keras-2.2 semantics; it is
cross-checked against an independent torch-autograd restatement in tests/test_oracle.py.
Row a10-i (EOS truncation) IS pinned: tests/test_oracle.py replays the reference's own known-answer
tests (reference tests/test_conv1d_text_vae.py:1544-1619) against `postprocess_reconstructed_word_vectors`.

All citations are file:line into /root/reference/conv1d_text_vae/conv1d_text_vae.py unless noted.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

TAU = 0.1                 # :1319
DROPOUT_RATE = 0.3        # :1243, :1254
KEEP_PROB = 1.0 - DROPOUT_RATE
L2_EPS = 1e-12            # tf.nn.l2_normalize epsilon used by K.l2_normalize (:1195)
XENT_EPS = 1e-7           # K.epsilon() used by K.categorical_crossentropy clip (:1198)
CLIPNORM = 10.0           # :1329
NADAM_LR = 0.002          # Keras-2.2 Nadam default; the estimator's `lr` is NOT used for the VAE (:1329)
NADAM_B1 = 0.9
NADAM_B2 = 0.999
NADAM_EPS = 1e-7          # K.epsilon()
NADAM_SCHEDULE_DECAY = 0.004
# Keras 2.2 BatchNormalization defaults (layers/normalization.py): axis -1, momentum 0.99, epsilon 1e-3, gamma ones,
# beta zeros, moving_mean zeros, moving_variance ones
BN_MOMENTUM = 0.99
BN_EPS = 1e-3
BN_WEIGHTS = ('gamma', 'beta', 'moving_mean', 'moving_variance')


# ---------------------------------------------------------------------------------------------
# geometry (follows __create_model :1224-1281)
# ---------------------------------------------------------------------------------------------
@dataclass
class VAEGeometry:
    input_text_size: int
    output_text_size: int
    input_vector_size: int
    output_vector_size: int
    n_filters: Tuple[int, ...]
    kernel_size: int
    latent_dim: int
    use_batch_norm: bool = False      # BatchNormalization after every conv / transposed conv (:1229-1230, :1259-1260)

    enc_len_conv: List[int] = field(default_factory=list)   # length after each valid conv
    enc_len_pool: List[int] = field(default_factory=list)   # length after each MaxPool1D(2)
    flat_size: int = 0
    n_times: int = 0          # n_times_of_decoder :1247-1250
    width: int = 0            # :1251
    dec_filters: List[int] = field(default_factory=list)    # :1256, :1264-1266
    dec_len_deconv: List[int] = field(default_factory=list)
    dec_len_up: List[int] = field(default_factory=list)
    crop: int = 0             # :1276

    def __post_init__(self):
        if isinstance(self.n_filters, int):
            self.n_filters = (self.n_filters,)
        self.n_filters = tuple(int(f) for f in self.n_filters)
        k = self.kernel_size
        L = self.input_text_size
        for _ in self.n_filters:
            L = L - k + 1                     # Conv1D 'valid' :1227, :1235
            if L < 1:
                raise ValueError('input_text_size is too small for this conv stack')
            self.enc_len_conv.append(L)
            L = L // 2                        # MaxPool1D(2) :1231, :1240
            if L < 1:
                raise ValueError('input_text_size is too small for this conv stack')
            self.enc_len_pool.append(L)
        self.flat_size = L * self.n_filters[-1]            # Flatten :1242
        n_t = int(math.ceil((self.output_text_size - k + 1) / 2.0))    # :1247
        for _ in range(len(self.n_filters) - 1):                       # :1248-1250
            n_t = int(math.ceil((n_t - k + 1) / 2.0))
        if n_t < 1:
            raise ValueError('output_text_size is too small for this conv stack')
        self.n_times = n_t
        self.width = max(10, int(math.ceil(self.latent_dim / float(n_t))))    # :1251
        self.dec_filters = [self.n_filters[-1]] + [self.n_filters[i] for i in
                                                   reversed(range(len(self.n_filters) - 1))]   # :1256, :1264
        M = n_t
        for _ in self.dec_filters:
            M = M + k - 1                     # Conv2DTranspose 'valid', stride 1 :1218
            self.dec_len_deconv.append(M)
            M = 2 * M                         # UpSampling1D(2) :1261, :1273
            self.dec_len_up.append(M)
        self.crop = max(0, M - self.output_text_size)      # :1276-1278
        if M < self.output_text_size:
            raise ValueError('decoder is shorter than output_text_size')

    def param_shapes(self) -> Dict[str, Tuple[int, ...]]:
        """Keras weight shapes keyed `<layer name>/<kernel|bias>` (layer names from :1227-1280)."""
        k = self.kernel_size
        shapes: Dict[str, Tuple[int, ...]] = {}
        c_in = self.input_vector_size
        for i, f in enumerate(self.n_filters):
            shapes['encoder_conv1d_%d/kernel' % (i + 1)] = (k, c_in, f)
            shapes['encoder_conv1d_%d/bias' % (i + 1)] = (f,)
            if self.use_batch_norm:
                for w in BN_WEIGHTS:
                    shapes['encoder_batchnorm_%d/%s' % (i + 1, w)] = (f,)
            c_in = f
        shapes['z_mean/kernel'] = (self.flat_size, self.latent_dim)
        shapes['z_mean/bias'] = (self.latent_dim,)
        shapes['z_log_var/kernel'] = (self.flat_size, self.latent_dim)
        shapes['z_log_var/bias'] = (self.latent_dim,)
        shapes['deconv_decoder_dense/kernel'] = (self.latent_dim, self.n_times * self.width)
        shapes['deconv_decoder_dense/bias'] = (self.n_times * self.width,)
        c_in = self.width
        for i, f in enumerate(self.dec_filters):
            shapes['deconv_decoder_%d_deconv1d_part2/kernel' % (i + 1)] = (k, 1, f, c_in)   # (kh, kw, out, in)
            shapes['deconv_decoder_%d_deconv1d_part2/bias' % (i + 1)] = (f,)
            if self.use_batch_norm:
                for w in BN_WEIGHTS:
                    shapes['deconv_decoder_batchnorm_%d/%s' % (i + 1, w)] = (f,)
            c_in = f
        shapes['deconv_decoder_embeddings/kernel'] = (k, c_in, self.output_vector_size)
        shapes['deconv_decoder_embeddings/bias'] = (self.output_vector_size,)
        return shapes

    def n_params(self) -> int:
        return int(sum(int(np.prod(s)) for s in self.param_shapes().values()))


def _glorot_limit(shape: Tuple[int, ...]) -> float:
    """Keras `glorot_uniform` fan computation (keras/initializers.py `_compute_fans`, channels_last)."""
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:
        receptive = int(np.prod(shape[:-2]))
        fan_in = shape[-2] * receptive
        fan_out = shape[-1] * receptive
    return math.sqrt(6.0 / (fan_in + fan_out))


def init_params(geom: VAEGeometry, seed: int = 0, dtype=np.float32) -> Dict[str, np.ndarray]:
    """Keras default initialisation: kernels glorot_uniform, biases zero."""
    rng = np.random.Generator(np.random.PCG64(seed))
    params = {}
    for name, shape in geom.param_shapes().items():
        if name.endswith(('/bias', '/beta', '/moving_mean')):
            params[name] = np.zeros(shape, dtype=dtype)
        elif name.endswith(('/gamma', '/moving_variance')):
            params[name] = np.ones(shape, dtype=dtype)
        else:
            lim = _glorot_limit(shape)
            params[name] = rng.uniform(-lim, lim, size=shape).astype(dtype)
    return params


# ---------------------------------------------------------------------------------------------
# layer primitives
# ---------------------------------------------------------------------------------------------
def _windows(x: np.ndarray, k: int) -> np.ndarray:
    """[B, L, C] -> [B, L-k+1, k*C] sliding windows (tap-major, channel-minor)."""
    B, L, C = x.shape
    s0, s1, s2 = x.strides
    w = np.lib.stride_tricks.as_strided(x, shape=(B, L - k + 1, k, C), strides=(s0, s1, s1, s2), writeable=False)
    return w.reshape(B, L - k + 1, k * C)


def conv1d_valid(x, W, b):
    """a1: out[b,t,f] = sum_{j,c} x[b,t+j,c] W[j,c,f] + bias[f]  (cross-correlation; Keras Conv1D 'valid')."""
    k, C, F = W.shape
    return _windows(np.ascontiguousarray(x), k) @ W.reshape(k * C, F) + b


def conv1d_valid_bwd(x, W, dout):
    k, C, F = W.shape
    B, L, _ = x.shape
    cols = _windows(np.ascontiguousarray(x), k)
    dW = (cols.reshape(-1, k * C).T @ dout.reshape(-1, F)).reshape(k, C, F)
    db = dout.sum(axis=(0, 1))
    dcols = (dout @ W.reshape(k * C, F).T).reshape(B, L - k + 1, k, C)
    dx = np.zeros_like(x)
    for j in range(k):
        dx[:, j:j + L - k + 1, :] += dcols[:, :, j, :]
    return dx, dW, db


def conv1d_same(x, W, b):
    """a7: Keras Conv1D 'same', stride 1: pad left (k-1)//2, right k//2."""
    k = W.shape[0]
    pl, pr = (k - 1) // 2, k // 2
    xp = np.pad(x, ((0, 0), (pl, pr), (0, 0)))
    return conv1d_valid(xp, W, b)


def conv1d_same_bwd(x, W, dout):
    k = W.shape[0]
    pl, pr = (k - 1) // 2, k // 2
    xp = np.pad(x, ((0, 0), (pl, pr), (0, 0)))
    dxp, dW, db = conv1d_valid_bwd(xp, W, dout)
    return dxp[:, pl:pl + x.shape[1], :], dW, db


def deconv1d_valid(x, Wt, b):
    """a6: Conv2DTranspose((k,1), stride 1, 'valid') used as 1-D (:1215-1222).
    Wt has the Keras layout (k, 1, F_out, C_in); out[b,t,f] = sum_{j,c} x[b,t-j,c] Wt[j,0,f,c] + bias[f]."""
    k, _, F, C = Wt.shape
    xp = np.pad(x, ((0, 0), (k - 1, k - 1), (0, 0)))
    Wc = np.ascontiguousarray(Wt[::-1, 0].transpose(0, 2, 1))     # [k, C, F], taps flipped
    return conv1d_valid(xp, Wc, b)


def deconv1d_valid_bwd(x, Wt, dout):
    k, _, F, C = Wt.shape
    xp = np.pad(x, ((0, 0), (k - 1, k - 1), (0, 0)))
    Wc = np.ascontiguousarray(Wt[::-1, 0].transpose(0, 2, 1))
    dxp, dWc, db = conv1d_valid_bwd(xp, Wc, dout)
    dWt = np.ascontiguousarray(dWc.transpose(0, 2, 1)[::-1])[:, None, :, :]
    return dxp[:, k - 1:k - 1 + x.shape[1], :], dWt, db


def maxpool2(x):
    """a2: MaxPool1D(2): out[b,t,f] = max(x[b,2t,f], x[b,2t+1,f]); odd tail dropped.
    Returns (out, sel) where sel==1 means the odd element was strictly larger (first max wins ties)."""
    B, L, F = x.shape
    Lo = L // 2
    a = x[:, 0:2 * Lo:2, :]
    c = x[:, 1:2 * Lo:2, :]
    sel = (c > a)
    return np.where(sel, c, a), sel


def maxpool2_bwd(dout, sel, L):
    B, Lo, F = dout.shape
    dx = np.zeros((B, L, F), dtype=dout.dtype)
    dx[:, 0:2 * Lo:2, :] = np.where(sel, 0, dout)
    dx[:, 1:2 * Lo:2, :] = np.where(sel, dout, 0)
    return dx


def upsample2(x):
    return np.repeat(x, 2, axis=1)        # UpSampling1D(2) == K.repeat_elements :1261


def upsample2_bwd(dout):
    B, L2, F = dout.shape
    return dout.reshape(B, L2 // 2, 2, F).sum(axis=2)


def batchnorm_fwd(x, params, name, training, cache=None, key=None):
    """Keras 2.2 BatchNormalization over the last axis of x [B, L, C] (:1229-1230, :1259-1260).
    training: statistics of THIS batch over (B, L) (biased variance) normalise it; the moving averages move by
    (1 - momentum) towards the batch mean and towards the batch variance scaled by n / (n - (1 + eps)) (Keras' "sample
    variance" correction).  Otherwise the moving averages normalise.  Returns (y, new moving averages or None)."""
    gamma, beta = params[name + '/gamma'], params[name + '/beta']
    if training:
        n = x.shape[0] * x.shape[1]
        mean = x.mean(axis=(0, 1))
        var = x.var(axis=(0, 1))
        rstd = 1.0 / np.sqrt(var + BN_EPS)
        xhat = (x - mean) * rstd
        moving = (params[name + '/moving_mean'] * BN_MOMENTUM + mean * (1.0 - BN_MOMENTUM),
                  params[name + '/moving_variance'] * BN_MOMENTUM + var * (n / (n - (1.0 + BN_EPS))) * (1.0 - BN_MOMENTUM))
    else:
        rstd = 1.0 / np.sqrt(params[name + '/moving_variance'] + BN_EPS)
        xhat = (x - params[name + '/moving_mean']) * rstd
        moving = None
    if cache is not None:
        cache[key + '_xhat'] = xhat
        cache[key + '_rstd'] = rstd
    return xhat * gamma + beta, moving


def batchnorm_bwd(dy, xhat, rstd, gamma):
    """Gradient of the training-mode normalisation: (dx, dgamma, dbeta)."""
    n = dy.shape[0] * dy.shape[1]
    dbeta = dy.sum(axis=(0, 1))
    dgamma = (dy * xhat).sum(axis=(0, 1))
    dx = (gamma * rstd) * (dy - dbeta / n - xhat * (dgamma / n))
    return dx, dgamma, dbeta


# ---------------------------------------------------------------------------------------------
# full model: forward / loss / backward
# ---------------------------------------------------------------------------------------------
def encoder_forward(geom: VAEGeometry, params, x, mask_e=None, cache: Optional[dict] = None, training: bool = False,
                    moving: Optional[dict] = None):
    """:1224-1245.  mask_e (bool [B, flat]) = dropout keep mask in training, None at inference (:1243).
    training / moving only matter with use_batch_norm: batch statistics are used and the new moving averages are put
    into `moving` (name -> array)."""
    h = x
    for i in range(len(geom.n_filters)):
        W = params['encoder_conv1d_%d/kernel' % (i + 1)]
        b = params['encoder_conv1d_%d/bias' % (i + 1)]
        pre = conv1d_valid(h, W, b)
        act = np.maximum(pre, 0)
        bn_in = act
        if geom.use_batch_norm:
            name = 'encoder_batchnorm_%d' % (i + 1)
            act_n, mv = batchnorm_fwd(act, params, name, training, cache, 'enc_bn_%d' % i)
            if moving is not None and mv is not None:
                moving[name + '/moving_mean'], moving[name + '/moving_variance'] = mv
            if cache is not None:
                cache['enc_relu_%d' % i] = bn_in
            act = act_n
        pooled, sel = maxpool2(act)
        if cache is not None:
            cache['enc_in_%d' % i] = h
            cache['enc_act_%d' % i] = act
            cache['enc_sel_%d' % i] = sel
        h = pooled
    flat = h.reshape(h.shape[0], -1)                        # Flatten: t-major, f-minor :1242
    if mask_e is not None:
        flat_d = flat * mask_e.astype(flat.dtype) / flat.dtype.type(KEEP_PROB)
    else:
        flat_d = flat
    z_mean = flat_d @ params['z_mean/kernel'] + params['z_mean/bias']              # :1244
    z_log_var = flat_d @ params['z_log_var/kernel'] + params['z_log_var/bias']     # :1245
    if cache is not None:
        cache['flat'] = flat
        cache['flat_d'] = flat_d
        cache['pool_shape'] = h.shape
    return z_mean, z_log_var


def decoder_forward(geom: VAEGeometry, params, z, mask_d=None, cache: Optional[dict] = None, training: bool = False,
                    moving: Optional[dict] = None):
    """DecoderForVAE :1252-1281."""
    if mask_d is not None:
        zd = z * mask_d.astype(z.dtype) / z.dtype.type(KEEP_PROB)      # :1254
    else:
        zd = z
    pre = zd @ params['deconv_decoder_dense/kernel'] + params['deconv_decoder_dense/bias']
    act = np.maximum(pre, 0)                                           # :1253
    d = act.reshape(z.shape[0], geom.n_times, geom.width)              # :1255
    if cache is not None:
        cache['zd'] = zd
        cache['dense_act'] = act
    for i in range(len(geom.dec_filters)):
        Wt = params['deconv_decoder_%d_deconv1d_part2/kernel' % (i + 1)]
        b = params['deconv_decoder_%d_deconv1d_part2/bias' % (i + 1)]
        pre = deconv1d_valid(d, Wt, b)
        a = np.maximum(pre, 0)
        if cache is not None:
            cache['dec_in_%d' % i] = d
            cache['dec_act_%d' % i] = a
        if geom.use_batch_norm:
            name = 'deconv_decoder_batchnorm_%d' % (i + 1)
            a, mv = batchnorm_fwd(a, params, name, training, cache, 'dec_bn_%d' % i)
            if moving is not None and mv is not None:
                moving[name + '/moving_mean'], moving[name + '/moving_variance'] = mv
        d = upsample2(a)
    full_len = d.shape[1]
    d = d[:, :geom.output_text_size, :]                                # Cropping1D((0, c)) :1276-1278
    y = conv1d_same(d, params['deconv_decoder_embeddings/kernel'], params['deconv_decoder_embeddings/bias'])
    if cache is not None:
        cache['final_in'] = d
        cache['dec_full_len'] = full_len
    return y


def sampling(z_mean, z_log_var, eps):
    """:1188-1191 -- note exp(z_log_var), NOT exp(0.5*z_log_var)."""
    return z_mean + np.exp(z_log_var) * eps


def vae_loss(y_pred, W_vocab, target_ids, z_mean, z_log_var, tau=TAU, cache: Optional[dict] = None):
    """vae_loss :1193-1203 followed by the Keras batch mean.

    W_vocab  [N, V_out]: rows of `output_word_vectors` (unit L2 norm), :1326 uses its transpose.
    target_ids [B, T_out] int: word id per step or -1 where the one-hot row is all-zero (SURVEY D9;
    identical maths to the dense one-hot target of :1372, mask = any(y_true != 0) :1199).
    Returns (loss scalar, per-sample xent [B], per-sample kl [B])."""
    dt = y_pred.dtype
    B, T, V = y_pred.shape
    sq = np.sum(y_pred * y_pred, axis=-1, keepdims=True)
    inv = 1.0 / np.sqrt(np.maximum(sq, dt.type(L2_EPS)))              # tf.nn.l2_normalize
    yhat = y_pred * inv
    s = (yhat.reshape(B * T, V) @ W_vocab.T) * dt.type(1.0 / tau)     # :1195
    s = s - s.max(axis=-1, keepdims=True)                             # softmax is shift invariant
    e = np.exp(s)
    p = e / e.sum(axis=-1, keepdims=True)                             # K.softmax :1194
    p = p / p.sum(axis=-1, keepdims=True)                             # categorical_crossentropy renormalises
    ids = target_ids.reshape(-1)
    mask = (ids >= 0)
    safe = np.where(mask, ids, 0)
    p_t = p[np.arange(B * T), safe]
    p_t_c = np.clip(p_t, XENT_EPS, 1.0 - XENT_EPS)                    # K.clip(output, eps, 1-eps)
    xent = np.where(mask, -np.log(p_t_c), 0).astype(dt)               # one-hot target => -log p[target]
    maskf = mask.astype(dt).reshape(B, T)
    xent_b = (xent.reshape(B, T) * maskf).sum(axis=-1) / maskf.sum(axis=-1)      # :1200-1201
    kl_b = np.mean(1 + z_log_var - np.square(z_mean) - np.exp(z_log_var), axis=-1)   # :1202 (mean, no 0.5)
    loss_b = xent_b - kl_b                                            # :1203
    if cache is not None:
        cache.update(dict(p=p, yhat=yhat, inv=inv, mask=mask, safe=safe, p_t=p_t, maskf=maskf))
    return loss_b.mean(), xent_b, kl_b


def vae_loss_bwd(y_pred, W_vocab, target_ids, z_mean, z_log_var, cache, tau=TAU):
    """Gradient of mean_b(vae_loss) w.r.t. y_pred, z_mean, z_log_var (W is a constant :1326)."""
    dt = y_pred.dtype
    B, T, V = y_pred.shape
    D = z_mean.shape[1]
    p, yhat, inv, mask, safe, p_t, maskf = (cache[k] for k in ('p', 'yhat', 'inv', 'mask', 'safe', 'p_t', 'maskf'))
    live = mask & (p_t > XENT_EPS) & (p_t < 1.0 - XENT_EPS)          # clip_by_value has zero grad outside
    coef = (live.astype(dt).reshape(B, T) / (maskf.sum(axis=-1, keepdims=True) * B)).reshape(-1, 1)
    ds = p.copy()
    ds[np.arange(B * T), safe] -= 1.0
    ds *= coef
    dyhat = (ds @ W_vocab) * dt.type(1.0 / tau)
    dyhat = dyhat.reshape(B, T, V)
    sq = np.sum(y_pred * y_pred, axis=-1, keepdims=True)
    normal = sq >= L2_EPS
    dy_n = (dyhat - yhat * np.sum(yhat * dyhat, axis=-1, keepdims=True)) * inv
    dy = np.where(normal, dy_n, dyhat * inv)
    dm = 2.0 * z_mean / (D * B)
    dlv = (np.exp(z_log_var) - 1.0) / (D * B)
    return dy.astype(dt), dm.astype(dt), dlv.astype(dt)


def train_forward_backward(geom: VAEGeometry, params, W_vocab, x, target_ids, eps,
                           mask_e=None, mask_d=None, frozen_encoder: bool = False, decisions: Optional[dict] = None):
    """One training forward+backward of FullVAE (:1320) with injected eps / dropout masks.
    Returns (loss, grads dict, activations dict).

    decisions (error attribution in the bf16 parity tests, not reference behaviour): the ReLU / max-pool choices of
    ANOTHER run of the same step -- `enc_sel_i` bool [B, L_pool, F] (odd element chosen), `enc_on_i` bool [B, L_pool, F]
    (chosen element > 0), `dec_on_i` bool [B, L, F], `dense_on` bool [B, n_t*width] -- replace this run's own choices in
    the backward pass, so that what is left of the difference between the two gradients is rounding, not flipped choices."""
    c: dict = {}
    moving: dict = {}
    z_mean, z_log_var = encoder_forward(geom, params, x, mask_e, c, training=True, moving=moving)
    z = sampling(z_mean, z_log_var, eps)
    y = decoder_forward(geom, params, z, mask_d, c, training=True, moving=moving)
    lc: dict = {}
    loss, xent_b, kl_b = vae_loss(y, W_vocab, target_ids, z_mean, z_log_var, cache=lc)
    dy, dm, dlv = vae_loss_bwd(y, W_vocab, target_ids, z_mean, z_log_var, lc)
    grads: Dict[str, np.ndarray] = {}
    dt = x.dtype
    # --- decoder backward ---
    dd, dW, db = conv1d_same_bwd(c['final_in'], params['deconv_decoder_embeddings/kernel'], dy)
    grads['deconv_decoder_embeddings/kernel'] = dW
    grads['deconv_decoder_embeddings/bias'] = db
    full = np.zeros((dd.shape[0], c['dec_full_len'], dd.shape[2]), dtype=dt)
    full[:, :geom.output_text_size, :] = dd                         # crop backward
    g = full
    for i in reversed(range(len(geom.dec_filters))):
        g = upsample2_bwd(g)
        if geom.use_batch_norm:
            bn = 'deconv_decoder_batchnorm_%d' % (i + 1)
            g, grads[bn + '/gamma'], grads[bn + '/beta'] = batchnorm_bwd(g, c['dec_bn_%d_xhat' % i], c['dec_bn_%d_rstd' % i],
                                                                         params[bn + '/gamma'])
        g = g * (decisions['dec_on_%d' % i] if decisions is not None else (c['dec_act_%d' % i] > 0))
        name = 'deconv_decoder_%d_deconv1d_part2' % (i + 1)
        g, dWt, db = deconv1d_valid_bwd(c['dec_in_%d' % i], params[name + '/kernel'], g)
        grads[name + '/kernel'] = dWt
        grads[name + '/bias'] = db
    g = g.reshape(g.shape[0], -1) * (decisions['dense_on'] if decisions is not None else (c['dense_act'] > 0))
    grads['deconv_decoder_dense/kernel'] = c['zd'].T @ g
    grads['deconv_decoder_dense/bias'] = g.sum(axis=0)
    dzd = g @ params['deconv_decoder_dense/kernel'].T
    dz = dzd * (mask_d.astype(dt) / dt.type(KEEP_PROB)) if mask_d is not None else dzd
    # --- sampling + KL ---
    dm_tot = dz + dm
    dlv_tot = dz * eps * np.exp(z_log_var) + dlv
    # --- heads ---
    flat_d = c['flat_d']
    grads['z_mean/kernel'] = flat_d.T @ dm_tot
    grads['z_mean/bias'] = dm_tot.sum(axis=0)
    grads['z_log_var/kernel'] = flat_d.T @ dlv_tot
    grads['z_log_var/bias'] = dlv_tot.sum(axis=0)
    dflat_d = dm_tot @ params['z_mean/kernel'].T + dlv_tot @ params['z_log_var/kernel'].T
    dflat = dflat_d * (mask_e.astype(dt) / dt.type(KEEP_PROB)) if mask_e is not None else dflat_d
    g = dflat.reshape(c['pool_shape'])
    for i in reversed(range(len(geom.n_filters))):
        act = c['enc_act_%d' % i]
        if geom.use_batch_norm:
            # pool backward, then the normalisation, then the ReLU of the convolution in front of it
            g = maxpool2_bwd(g, c['enc_sel_%d' % i], act.shape[1])
            bn = 'encoder_batchnorm_%d' % (i + 1)
            g, grads[bn + '/gamma'], grads[bn + '/beta'] = batchnorm_bwd(g, c['enc_bn_%d_xhat' % i], c['enc_bn_%d_rstd' % i],
                                                                         params[bn + '/gamma'])
            g = g * (c['enc_relu_%d' % i] > 0)
        elif decisions is not None:
            g = maxpool2_bwd(g * decisions['enc_on_%d' % i], decisions['enc_sel_%d' % i], act.shape[1])
        else:
            g = maxpool2_bwd(g, c['enc_sel_%d' % i], act.shape[1])
            g = g * (act > 0)
        name = 'encoder_conv1d_%d' % (i + 1)
        g, dW, db = conv1d_valid_bwd(c['enc_in_%d' % i], params[name + '/kernel'], g)
        grads[name + '/kernel'] = dW
        grads[name + '/bias'] = db
    if frozen_encoder:                                               # warm_start: trainable=False :1228, :1244
        for name in list(grads):
            if name.startswith('encoder_conv1d_') or name.startswith('z_mean') or name.startswith('z_log_var'):
                grads[name] = np.zeros_like(grads[name])
    acts = dict(z_mean=z_mean, z_log_var=z_log_var, z=z, y_pred=y, xent_b=xent_b, kl_b=kl_b, dy=dy,
                flat=c['flat'], final_in=c['final_in'], moving=moving)
    return float(loss), {k: v.astype(dt) for k, v in grads.items()}, acts


def eval_loss(geom, params, W_vocab, x, target_ids, eps):
    """Validation-mode loss: dropout off (Keras learning_phase 0), sampling Lambda still draws eps."""
    z_mean, z_log_var = encoder_forward(geom, params, x)
    z = sampling(z_mean, z_log_var, eps)
    y = decoder_forward(geom, params, z)
    loss, _, _ = vae_loss(y, W_vocab, target_ids, z_mean, z_log_var)
    return float(loss)


# ---------------------------------------------------------------------------------------------
# a9: Nadam(clipnorm=10) -- standalone Keras 2.2 `optimizers.Nadam.get_updates` + `clip_norm`
# ---------------------------------------------------------------------------------------------
@dataclass
class NadamState:
    iterations: int = 0
    m_schedule: float = 1.0
    m: Dict[str, np.ndarray] = field(default_factory=dict)
    v: Dict[str, np.ndarray] = field(default_factory=dict)


def global_norm(grads: Dict[str, np.ndarray]) -> float:
    return math.sqrt(float(sum(np.sum(np.square(g.astype(np.float64))) for g in grads.values())))


def nadam_schedule(t: int, m_schedule: float):
    """Scalars of step t (1-based): returns (mu_t, mu_t1, m_schedule_new, m_schedule_next)."""
    mu_t = NADAM_B1 * (1.0 - 0.5 * math.pow(0.96, t * NADAM_SCHEDULE_DECAY))
    mu_t1 = NADAM_B1 * (1.0 - 0.5 * math.pow(0.96, (t + 1) * NADAM_SCHEDULE_DECAY))
    m_new = m_schedule * mu_t
    m_next = m_schedule * mu_t * mu_t1
    return mu_t, mu_t1, m_new, m_next


def nadam_step(params: Dict[str, np.ndarray], grads: Dict[str, np.ndarray], state: NadamState,
               lr: float = NADAM_LR, clipnorm: float = CLIPNORM) -> float:
    """In-place update; returns the un-clipped global gradient norm."""
    norm = global_norm(grads)
    scale = (clipnorm / norm) if (clipnorm > 0 and norm >= clipnorm) else 1.0     # keras.optimizers.clip_norm
    t = state.iterations + 1
    mu_t, mu_t1, m_new, m_next = nadam_schedule(t, state.m_schedule)
    for name, p in params.items():
        if name not in grads:          # BatchNormalization moving averages: not trainable, moved by the forward pass
            continue
        dt = p.dtype
        g = grads[name] * dt.type(scale)
        m = state.m.get(name)
        v = state.v.get(name)
        if m is None:
            m = np.zeros_like(p)
            v = np.zeros_like(p)
        g_prime = g / dt.type(1.0 - m_new)
        m_t = dt.type(NADAM_B1) * m + dt.type(1.0 - NADAM_B1) * g
        m_t_prime = m_t / dt.type(1.0 - m_next)
        v_t = dt.type(NADAM_B2) * v + dt.type(1.0 - NADAM_B2) * np.square(g)
        v_t_prime = v_t / dt.type(1.0 - math.pow(NADAM_B2, t))
        m_bar = dt.type(1.0 - mu_t) * g_prime + dt.type(mu_t1) * m_t_prime
        p -= dt.type(lr) * m_bar / (np.sqrt(v_t_prime) + dt.type(NADAM_EPS))
        state.m[name] = m_t
        state.v[name] = v_t
    state.iterations = t
    state.m_schedule = m_new
    return norm


# ---------------------------------------------------------------------------------------------
# a10-i: EOS truncation / renormalisation (:542-565), vectorised restatement
# ---------------------------------------------------------------------------------------------
def postprocess_reconstructed_word_vectors(word_vectors: np.ndarray) -> np.ndarray:
    if len(word_vectors.shape) != 3:
        raise ValueError('The `word_vectors` parameter is wrong! Expected 3-D array, got {0}-D array!'.format(
            len(word_vectors.shape)))
    B, T, V = word_vectors.shape
    out = np.zeros((B, T, V - 1), dtype=word_vectors.dtype)
    is_eos = (np.argmax(word_vectors, axis=-1) == (V - 1))          # first maximum wins, as np.argmax :554
    for b in range(B):
        hits = np.nonzero(is_eos[b])[0]
        end = T if hits.size == 0 else max(int(hits[0]), 1)           # :556
        rows = word_vectors[b, :end, :V - 1]
        norms = np.linalg.norm(rows, axis=-1, keepdims=True)          # no zero guard in the reference :563-564
        with np.errstate(divide='ignore', invalid='ignore'):
            out[b, :end] = rows / norms
    return out


def eos_end_index(word_vectors: np.ndarray) -> np.ndarray:
    B, T, V = word_vectors.shape
    is_eos = (np.argmax(word_vectors, axis=-1) == (V - 1))
    end = np.full((B,), T, dtype=np.int32)
    for b in range(B):
        hits = np.nonzero(is_eos[b])[0]
        if hits.size:
            end[b] = max(int(hits[0]), 1)
    return end


# ---------------------------------------------------------------------------------------------
# a10-ii: nearest word (gensim `similar_by_vector` arithmetic, :587) as a batched GEMM + arg-max
# ---------------------------------------------------------------------------------------------
def vocab_argmax(queries: np.ndarray, vocab_norm: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """queries [M, K], vocab_norm [N, K] (unit rows). Returns (best score f32 [M], best index i32 [M]);
    ties -> lowest index (np.argmax / argsort first occurrence)."""
    s = queries @ vocab_norm.T
    idx = np.argmax(s, axis=-1)
    return s[np.arange(s.shape[0]), idx].astype(np.float32), idx.astype(np.int32)


def cosine_distance(u, v):
    """scipy.spatial.distance.cosine."""
    uv = float(np.dot(u, v))
    uu = float(np.dot(u, u))
    vv = float(np.dot(v, v))
    return 1.0 - uv / math.sqrt(uu * vv)


def find_best_words(word_vector: np.ndarray, emb_matrix_norm: np.ndarray, words: Sequence[str], emb_raw: np.ndarray,
                    n: int, special_symbols: Optional[tuple] = None):
    """Restatement of Conv1dTextVAE.find_best_words (:580-642) over an explicit embedding table.
    emb_matrix_norm [Nw, E] = gensim `vectors_norm`; emb_raw [Nw, E] = gensim `vectors` (emb[word])."""
    E = emb_raw.shape[1]
    vector_size = E + 2 + (0 if special_symbols is None else len(special_symbols))
    norm_value = np.linalg.norm(word_vector[:E])
    if norm_value < 1e-7:
        norm_value = 1.0
    q = word_vector[:E] / norm_value
    dists = emb_matrix_norm @ q
    order = np.argsort(-dists, kind='stable')[:n]
    res = [(words[i], float(dists[i])) for i in order]
    best_vector = np.zeros((vector_size,), dtype=np.float32)
    best_vector[:E] = emb_raw[order[0]]
    nv = np.linalg.norm(best_vector)
    if nv > 0.0:
        best_vector /= nv
    end_v = np.zeros((vector_size,), dtype=np.float32)
    end_v[vector_size - 1] = 1.0
    unk_v = np.zeros((vector_size,), dtype=np.float32)
    unk_v[vector_size - 2] = 1.0
    if special_symbols is not None and len(special_symbols) > 0:
        d_special = np.zeros((len(special_symbols),), dtype=np.float32)
        for i in range(len(special_symbols)):
            sv = np.zeros((vector_size,), dtype=np.float32)
            sv[E + i] = 1.0
            d_special[i] = cosine_distance(word_vector, sv)
        special_idx = int(d_special.argmin())
    else:
        special_idx = -1
        d_special = None
    d_end = cosine_distance(word_vector, end_v)
    d_unk = cosine_distance(word_vector, unk_v)
    d_best = cosine_distance(word_vector, best_vector)
    if d_end < d_unk:
        if d_end < d_best:
            if special_idx >= 0:
                if d_end < d_special[special_idx]:
                    res = None
                else:
                    res = [(special_symbols[special_idx], d_special[special_idx])]
            else:
                res = None
    else:
        if d_unk < d_best:
            if special_idx >= 0:
                if d_unk < d_special[special_idx]:
                    res = []
                else:
                    res = [(special_symbols[special_idx], d_special[special_idx])]
            else:
                res = []
        else:
            if special_idx >= 0:
                if d_special[special_idx] < d_best:
                    res = [(special_symbols[special_idx], d_special[special_idx])]
            else:
                vecs = np.zeros((len(order), vector_size), dtype=np.float32)
                for i, wi in enumerate(order):
                    vecs[i, :E] = emb_raw[wi]
                    nv = np.linalg.norm(vecs[i])
                    if nv > 0.0:
                        vecs[i] /= nv
                res = [(res[i][0], cosine_distance(word_vector, vecs[i])) for i in range(len(order))]
                res.sort(key=lambda it: (it[1], it[0]))
    return res


# ---------------------------------------------------------------------------------------------
# counter-based RNG (Philox4x32-10), the build's replacement for TF's unseeded RNG
# ---------------------------------------------------------------------------------------------
_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = np.uint32(0x9E3779B9)
_PHILOX_W1 = np.uint32(0xBB67AE85)


def philox4x32(counter: np.ndarray, key: Tuple[int, int]) -> np.ndarray:
    """counter uint32 [n, 4] -> uint32 [n, 4] (10 rounds)."""
    c = counter.astype(np.uint32).copy()
    k0 = np.uint32(key[0] & 0xFFFFFFFF)
    k1 = np.uint32(key[1] & 0xFFFFFFFF)
    for _ in range(10):
        p0 = c[:, 0].astype(np.uint64) * _PHILOX_M0
        p1 = c[:, 2].astype(np.uint64) * _PHILOX_M1
        hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
        lo0 = (p0 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
        lo1 = (p1 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        n0 = hi1 ^ c[:, 1] ^ k0
        n1 = lo1
        n2 = hi0 ^ c[:, 3] ^ k1
        n3 = lo0
        c = np.stack([n0, n1, n2, n3], axis=1).astype(np.uint32)
        with np.errstate(over='ignore'):
            k0 = np.uint32((int(k0) + int(_PHILOX_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_PHILOX_W1)) & 0xFFFFFFFF)
    return c


def _philox_words(n: int, seed: int, stream: int, offset: int) -> np.ndarray:
    """One uint32 word per element index e in [offset, offset+n): word (e & 3) of block (e >> 2)."""
    e = np.arange(offset, offset + n, dtype=np.uint64)
    blk = e >> np.uint64(2)
    ctr = np.zeros((n, 4), dtype=np.uint32)
    ctr[:, 0] = (blk & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ctr[:, 1] = (blk >> np.uint64(32)).astype(np.uint32)
    ctr[:, 2] = np.uint32(stream & 0xFFFFFFFF)
    ctr[:, 3] = np.uint32((stream >> 32) & 0xFFFFFFFF)
    out = philox4x32(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    return out[np.arange(n), (e & np.uint64(3)).astype(np.int64)]


def philox_bernoulli(n: int, keep_prob: float, seed: int, stream: int, offset: int = 0) -> np.ndarray:
    """keep mask uint8 [n]: keep iff (word >> 8) * 2^-24 < keep_prob, compared in integers."""
    w = _philox_words(n, seed, stream, offset)
    thr = np.uint32(int(keep_prob * 16777216.0))
    return ((w >> np.uint32(8)) < thr).astype(np.uint8)


def philox_normal(n: int, seed: int, stream: int, offset: int = 0) -> np.ndarray:
    """N(0,1) float32 [n] by Box-Muller over the element's own block: elements 4b..4b+3 use the block's
    words (w0,w1) -> (r cos, r sin) and (w2,w3) -> (r cos, r sin)."""
    e = np.arange(offset, offset + n, dtype=np.uint64)
    blk = e >> np.uint64(2)
    ctr = np.zeros((n, 4), dtype=np.uint32)
    ctr[:, 0] = (blk & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ctr[:, 1] = (blk >> np.uint64(32)).astype(np.uint32)
    ctr[:, 2] = np.uint32(stream & 0xFFFFFFFF)
    ctr[:, 3] = np.uint32((stream >> 32) & 0xFFFFFFFF)
    w = philox4x32(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    lane = (e & np.uint64(3)).astype(np.int64)
    pair = lane >> 1
    wa = w[np.arange(n), 2 * pair]
    wb = w[np.arange(n), 2 * pair + 1]
    u1 = ((wa >> np.uint32(8)).astype(np.float32) + np.float32(1.0)) * np.float32(1.0 / 16777216.0)   # (0,1]
    u2 = (wb >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)                       # [0,1)
    r = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
    ang = (np.float32(2.0 * math.pi) * u2).astype(np.float32)
    return np.where((lane & 1) == 0, r * np.cos(ang), r * np.sin(ang)).astype(np.float32)


# ---------------------------------------------------------------------------------------------
# synthetic workload (SURVEY.md 8d / BASELINE.md): shared by tests, smoke and the bench's cpu_baseline leg
# ---------------------------------------------------------------------------------------------
def synthetic_vocab(n_words: int, emb: int, seed: int = 1234, dtype=np.float32) -> np.ndarray:
    """[n_words + 2, emb + 2]: unit-norm N(0,1) rows, then the unknown-word one-hot row (slot emb) and the
    end-of-sentence one-hot row (last slot), exactly as prepare_vocabulary_and_word_vectors (:1051-1061)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    W = np.zeros((n_words + 2, emb + 2), dtype=dtype)
    v = rng.standard_normal((n_words, emb)).astype(dtype)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    W[:n_words, :emb] = v
    W[n_words, emb] = 1.0
    W[n_words + 1, emb + 1] = 1.0
    return W


def synthetic_batch(W_in: np.ndarray, W_out_rows: int, batch: int, t_in: int, t_out: int, seed: int = 1234):
    """Input [B, T_in, V] and target ids [B, T_out] built as SequenceForVAE.__getitem__ (:1366-1405):
    word rows for t < len, the EOS one-hot row at t == len, zeros after; target id per step, EOS id at
    t == len, -1 (all-zero one-hot) after."""
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    n_in = W_in.shape[0]
    eos_in = n_in - 1
    eos_out = W_out_rows - 1
    x = np.zeros((batch, t_in, W_in.shape[1]), dtype=W_in.dtype)
    ids = np.full((batch, t_out), -1, dtype=np.int32)
    for b in range(batch):
        li = int(rng.integers(3, t_in))
        wi = rng.integers(0, n_in - 2, size=li)
        x[b, :li] = W_in[wi]
        x[b, li] = W_in[eos_in]
        lo = int(rng.integers(3, t_out))
        ids[b, :lo] = rng.integers(0, W_out_rows - 2, size=lo)
        ids[b, lo] = eos_out
    return x, ids
