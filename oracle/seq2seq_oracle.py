"""TEST INFRASTRUCTURE ONLY.  NumPy restatement of the forward arithmetic of the reference's stage-2 generator
(conv1d_text_vae.py:1205-1213 attention, :1282-1318 GRUs + dense, Keras 2.2 defaults: gate order z, r, h;
reset_after=False; hard_sigmoid recurrent activation; Masking(0.0) carries state over all-zero steps; masked
categorical cross-entropy = sum(loss*mask)/sum(mask)).  Parity unpinned against Keras itself (not installable here);
it pins the torch implementation in conv1d_text_vae_amd/seq2seq.py against an independent loop-level statement."""
import numpy as np


def hard_sigmoid(x):
    return np.clip(0.2 * x + 0.5, 0.0, 1.0)


def attention(x, K):
    # Permute -> Dense(T, softmax, no bias) over the last axis -> mean over features -> scale the rows (:1205-1213)
    a = np.einsum('btf,ts->bfs', x, K)
    a = np.exp(a - a.max(axis=-1, keepdims=True))
    a /= a.sum(axis=-1, keepdims=True)
    return x * a.mean(axis=1)[:, :, None]


def gru(x, h0, W, U, b, masked):
    B, T, _ = x.shape
    u = U.shape[0]
    h = h0.copy()
    seq = np.zeros((B, T, u))
    for bi in range(B):
        for t in range(T):
            xt = x[bi, t]
            if masked and not np.any(xt != 0):
                seq[bi, t] = h[bi]
                continue
            z = hard_sigmoid(xt @ W[:, :u] + b[:u] + h[bi] @ U[:, :u])
            r = hard_sigmoid(xt @ W[:, u:2 * u] + b[u:2 * u] + h[bi] @ U[:, u:2 * u])
            hh = np.tanh(xt @ W[:, 2 * u:] + b[2 * u:] + (r * h[bi]) @ U[:, 2 * u:])
            h[bi] = z * h[bi] + (1.0 - z) * hh
            seq[bi, t] = h[bi]
    return seq, h


def seq2seq_loss(p, words, dec_in, dec_target, use_attention=True):
    x = attention(words, p['attention_aux3/kernel']) if use_attention else words
    u = p['seq2seq_encoder_gru/recurrent_kernel'].shape[0]
    _, state = gru(x, np.zeros((words.shape[0], u)), p['seq2seq_encoder_gru/kernel'],
                   p['seq2seq_encoder_gru/recurrent_kernel'], p['seq2seq_encoder_gru/bias'], True)
    seq, _ = gru(dec_in, state, p['seq2seq_decoder_gru/kernel'], p['seq2seq_decoder_gru/recurrent_kernel'],
                 p['seq2seq_decoder_gru/bias'], True)
    logits = seq @ p['seq2seq_decoder_dense/kernel'] + p['seq2seq_decoder_dense/bias']
    e = np.exp(logits - logits.max(axis=-1, keepdims=True))
    prob = e / e.sum(axis=-1, keepdims=True)
    prob = np.clip(prob / prob.sum(axis=-1, keepdims=True), 1e-7, 1.0 - 1e-7)
    xent = -(dec_target * np.log(prob)).sum(axis=-1)
    mask = np.any(dec_in != 0, axis=-1).astype(np.float64)
    return float((xent * mask).sum() / mask.sum()), state, prob
