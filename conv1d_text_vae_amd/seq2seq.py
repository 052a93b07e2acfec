"""Stage 2 of the reference: the character-level GRU generator that turns reconstructed word vectors into text
(graph: conv1d_text_vae.py:1205-1213, :1282-1318, :1327-1331; training: :342-376; greedy decoding: :442-470).

SURVEY.md 8(f) row 1: this stage is recurrent, tiny and latency-bound, so it is NOT part of the hand-written HIP hot
path; it runs on stock PyTorch-ROCm tensor ops (device = the estimator's GPU; `cpu` is accepted so that the Keras
semantics restated here can be unit-tested without a GPU).  What is restated from Keras 2.2 (not in the tree):

  * attention block (:1205-1213): a = softmax(permute(x) . K) over the last axis, mean over the feature axis, rows of x
    scaled by a[b, t];
  * `Masking(0.0)`: a step whose features are all zero is skipped (state and output carried over);
  * `GRU(units, dropout=0.5, recurrent_dropout=0.3)` defaults: gate order z, r, h; `reset_after=False`
    (hh = tanh(x W_h + (r*h) U_h + b_h)); recurrent activation hard_sigmoid = clip(0.2 x + 0.5, 0, 1); implementation 1:
    one input-dropout mask and one recurrent-dropout mask PER GATE, constant over time, training phase only;
  * loss: categorical cross-entropy on the softmax output with Keras' renormalise + clip(1e-7), weighted by the output
    mask as `weighted_masked_objective` does: sum(loss * mask) / sum(mask);
  * optimizer: RMSprop(lr=self.lr, rho=0.9, epsilon=1e-7) with global-norm clipping at 10 (standalone Keras `clip_norm`).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

CLIPNORM = 10.0
RMS_RHO = 0.9
RMS_EPS = 1e-7
GRU_DROPOUT = 0.5
GRU_RECURRENT_DROPOUT = 0.3


def hard_sigmoid(x: torch.Tensor) -> torch.Tensor:
    return torch.clamp(0.2 * x + 0.5, 0.0, 1.0)


def _glorot(rng: np.random.Generator, shape) -> np.ndarray:
    lim = np.sqrt(6.0 / (shape[0] + shape[1]))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def _orthogonal(rng: np.random.Generator, shape) -> np.ndarray:
    """Keras `Orthogonal` initializer (gain 1): SVD of a normal matrix, u or v whichever has the asked shape."""
    a = rng.standard_normal((shape[0], shape[1]))
    u, _, v = np.linalg.svd(a, full_matrices=False)
    q = u if u.shape == tuple(shape) else v
    return q.reshape(shape).astype(np.float32)


class Seq2SeqGenerator:
    ENCODER_KEYS = ('attention_aux3/kernel', 'seq2seq_encoder_gru/kernel', 'seq2seq_encoder_gru/recurrent_kernel',
                    'seq2seq_encoder_gru/bias')
    DECODER_KEYS = ('seq2seq_decoder_gru/kernel', 'seq2seq_decoder_gru/recurrent_kernel', 'seq2seq_decoder_gru/bias',
                    'seq2seq_decoder_dense/kernel', 'seq2seq_decoder_dense/bias')

    def __init__(self, text_size: int, feature_size: int, n_chars: int, units: int, use_attention: bool = True,
                 lr: float = 0.001, device: str = 'cuda:0', seed: int = 0):
        self.T, self.F, self.C, self.U = int(text_size), int(feature_size), int(n_chars), int(units)
        self.use_attention = bool(use_attention)
        self.lr = float(lr)
        self.dev = torch.device(device)
        rng = np.random.Generator(np.random.PCG64(seed))
        w: Dict[str, np.ndarray] = {}
        if self.use_attention:
            w['attention_aux3/kernel'] = _glorot(rng, (self.T, self.T))
        for name, in_dim in (('seq2seq_encoder_gru', self.F), ('seq2seq_decoder_gru', self.C)):
            w[name + '/kernel'] = _glorot(rng, (in_dim, 3 * self.U))
            w[name + '/recurrent_kernel'] = np.concatenate([_orthogonal(rng, (self.U, self.U)) for _ in range(3)], axis=1)
            w[name + '/bias'] = np.zeros((3 * self.U,), dtype=np.float32)
        w['seq2seq_decoder_dense/kernel'] = _glorot(rng, (self.U, self.C))
        w['seq2seq_decoder_dense/bias'] = np.zeros((self.C,), dtype=np.float32)
        self.p: Dict[str, torch.Tensor] = {k: torch.from_numpy(v).to(self.dev).requires_grad_(True) for k, v in w.items()}
        self.rms: Dict[str, torch.Tensor] = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.gen = torch.Generator(device=self.dev)
        self.gen.manual_seed(int(seed) + 17)

    # ---- weights ---------------------------------------------------------------------------------------
    def get_weights(self, keys=None) -> Dict[str, np.ndarray]:
        return {k: v.detach().cpu().numpy().copy() for k, v in self.p.items() if keys is None or k in keys}

    def set_weights(self, w: Dict[str, np.ndarray]):
        for k, v in w.items():
            if k not in self.p:
                raise ValueError('unknown weight `{0}`'.format(k))
            if tuple(v.shape) != tuple(self.p[k].shape):
                raise ValueError('weight `{0}` has shape {1}, expected {2}'.format(k, v.shape, tuple(self.p[k].shape)))
            with torch.no_grad():
                self.p[k].copy_(torch.as_tensor(np.asarray(v, dtype=np.float32)))

    # ---- graph -----------------------------------------------------------------------------------------
    def attention(self, x: torch.Tensor) -> torch.Tensor:
        """x [B, T, F] -> x * a[b, t]; a = mean_f softmax_t'(sum_t x[b, t, f] K[t, t'])   (:1205-1213)"""
        a = torch.softmax(torch.matmul(x.transpose(1, 2), self.p['attention_aux3/kernel']), dim=-1)      # [B, F, T]
        return x * a.mean(dim=1).unsqueeze(-1)

    def _gru(self, name: str, x: torch.Tensor, h0: torch.Tensor, masked: bool, train: bool, return_sequences: bool):
        W, U, b = self.p[name + '/kernel'], self.p[name + '/recurrent_kernel'], self.p[name + '/bias']
        B, T, _ = x.shape
        u = self.U
        mask = (x != 0).any(dim=-1) if masked else None                  # Masking(mask_value=0.0)
        if train:
            def drop(shape, rate):
                keep = (torch.rand(shape, device=x.device, generator=self.gen) >= rate).to(x.dtype)
                return keep / (1.0 - rate)
            dp = [drop((B, 1, x.shape[2]), GRU_DROPOUT) for _ in range(3)]
            rdp = [drop((B, u), GRU_RECURRENT_DROPOUT) for _ in range(3)]
            xz = torch.matmul(x * dp[0], W[:, :u]) + b[:u]
            xr = torch.matmul(x * dp[1], W[:, u:2 * u]) + b[u:2 * u]
            xh = torch.matmul(x * dp[2], W[:, 2 * u:]) + b[2 * u:]
        else:
            rdp = None
            xw = torch.matmul(x, W) + b
            xz, xr, xh = xw[..., :u], xw[..., u:2 * u], xw[..., 2 * u:]
        Uz, Ur, Uh = U[:, :u], U[:, u:2 * u], U[:, 2 * u:]
        h = h0
        outs = []
        for t in range(T):
            hz, hr, hh_in = (h * rdp[0], h * rdp[1], h * rdp[2]) if rdp is not None else (h, h, h)
            z = hard_sigmoid(xz[:, t] + hz @ Uz)
            r = hard_sigmoid(xr[:, t] + hr @ Ur)
            hh = torch.tanh(xh[:, t] + (r * hh_in) @ Uh)
            h_new = z * h + (1.0 - z) * hh
            if mask is not None:
                h_new = torch.where(mask[:, t].unsqueeze(-1), h_new, h)
            h = h_new
            if return_sequences:
                outs.append(h)
        return (torch.stack(outs, dim=1) if return_sequences else None), h, mask

    def encode(self, word_vectors: torch.Tensor, train: bool = False) -> torch.Tensor:
        """generator_encoder_ (:1309): reconstructed word vectors [B, T, F] -> GRU state [B, units]."""
        x = self.attention(word_vectors) if self.use_attention else word_vectors
        h0 = torch.zeros((x.shape[0], self.U), dtype=x.dtype, device=x.device)
        _, h, _ = self._gru('seq2seq_encoder_gru', x, h0, masked=True, train=train, return_sequences=False)
        return h

    def decode_step(self, chars_onehot: torch.Tensor, state: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """generator_decoder_ (:1311-1318): ([B, L, C] one-hot, [B, units]) -> (softmax [B, L, C], new state)."""
        seq, h, _ = self._gru('seq2seq_decoder_gru', chars_onehot, state, masked=False, train=False, return_sequences=True)
        logits = seq @ self.p['seq2seq_decoder_dense/kernel'] + self.p['seq2seq_decoder_dense/bias']
        return torch.softmax(logits, dim=-1), h

    def loss(self, word_vectors: torch.Tensor, dec_in: torch.Tensor, dec_target: torch.Tensor, train: bool) -> torch.Tensor:
        """seq2seq_model_for_training (:1327) with Keras' masked categorical cross-entropy."""
        state = self.encode(word_vectors, train=train)
        seq, _, mask = self._gru('seq2seq_decoder_gru', dec_in, state, masked=True, train=train, return_sequences=True)
        p = torch.softmax(seq @ self.p['seq2seq_decoder_dense/kernel'] + self.p['seq2seq_decoder_dense/bias'], dim=-1)
        p = p / p.sum(dim=-1, keepdim=True)
        p = torch.clamp(p, 1e-7, 1.0 - 1e-7)
        xent = -(dec_target * torch.log(p)).sum(dim=-1)                                  # [B, L]
        m = mask.to(xent.dtype)
        return (xent * m).sum() / m.sum()

    # ---- training --------------------------------------------------------------------------------------
    def train_batch(self, word_vectors: torch.Tensor, dec_in: torch.Tensor, dec_target: torch.Tensor) -> torch.Tensor:
        for v in self.p.values():
            v.grad = None
        loss = self.loss(word_vectors, dec_in, dec_target, train=True)
        loss.backward()
        with torch.no_grad():
            sq = torch.zeros((), dtype=torch.float32, device=self.dev)
            for v in self.p.values():
                if v.grad is not None:
                    sq = sq + (v.grad * v.grad).sum()
            norm = torch.sqrt(sq)
            scale = torch.where(norm >= CLIPNORM, CLIPNORM / norm, torch.ones_like(norm))    # keras clip_norm
            for k, v in self.p.items():
                if v.grad is None:
                    continue
                g = v.grad * scale
                a = self.rms[k]
                a.mul_(RMS_RHO).add_((1.0 - RMS_RHO) * g * g)
                v.sub_(self.lr * g / (torch.sqrt(a) + RMS_EPS))
        return loss.detach()

    @torch.no_grad()
    def eval_batch(self, word_vectors, dec_in, dec_target) -> torch.Tensor:
        return self.loss(word_vectors, dec_in, dec_target, train=False)

    # ---- greedy decoding (:442-470) with the per-character bookkeeping kept on the device -----------------
    @torch.no_grad()
    def greedy_decode(self, state: torch.Tensor, bos_idx: int, eos_idx: int, max_chars: int) -> np.ndarray:
        """Returns the sampled character ids [B, <= max_chars + 1]; a row is meaningful up to and including its first
        `eos_idx` (the reference stops a text when it samples <EOS> or when it holds more than `max_chars` characters).
        One host synchronisation per call instead of one per character."""
        B = state.shape[0]
        cur = torch.full((B,), int(bos_idx), dtype=torch.int64, device=state.device)
        done = torch.zeros((B,), dtype=torch.bool, device=state.device)
        out = torch.full((B, max_chars + 1), int(eos_idx), dtype=torch.int64, device=state.device)
        eye = torch.eye(self.C, dtype=state.dtype, device=state.device)
        check_every = 16
        for step in range(max_chars + 1):
            probs, state = self.decode_step(eye[cur].unsqueeze(1), state)
            cur = torch.argmax(probs[:, -1, :], dim=1)                # first maximum wins, as np.argmax (:457)
            out[:, step] = torch.where(done, out[:, step], cur)
            done = done | (cur == eos_idx)
            if (step + 1) % check_every == 0 and bool(done.all().item()):
                break
        return out.cpu().numpy()

    # ---- Keras-style model facades ------------------------------------------------------------------------
    def encoder_model(self) -> 'GeneratorEncoderModel':
        return GeneratorEncoderModel(self)

    def decoder_model(self) -> 'GeneratorDecoderModel':
        return GeneratorDecoderModel(self)


class GeneratorEncoderModel:
    """`generator_encoder_`: predict_on_batch(np [B, T, F]) -> np [B, units]   (:442)"""

    def __init__(self, gen: Seq2SeqGenerator):
        self.gen = gen

    @torch.no_grad()
    def predict_on_batch(self, x) -> np.ndarray:
        xt = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x, dtype=np.float32))
        return self.gen.encode(xt.to(self.gen.dev), train=False).cpu().numpy()

    def get_weights(self):
        return self.gen.get_weights(Seq2SeqGenerator.ENCODER_KEYS)


class GeneratorDecoderModel:
    """`generator_decoder_`: predict_on_batch([np [B, L, C], np [B, units]]) -> [np [B, L, C], np [B, units]]   (:456)"""

    def __init__(self, gen: Seq2SeqGenerator):
        self.gen = gen

    @torch.no_grad()
    def predict_on_batch(self, inputs) -> List[np.ndarray]:
        seq, state = inputs
        st = torch.as_tensor(np.asarray(seq, dtype=np.float32)).to(self.gen.dev)
        hs = torch.as_tensor(np.asarray(state, dtype=np.float32)).to(self.gen.dev)
        probs, h = self.gen.decode_step(st, hs)
        return [probs.cpu().numpy(), h.cpu().numpy()]

    def get_weights(self):
        return self.gen.get_weights(Seq2SeqGenerator.DECODER_KEYS)
