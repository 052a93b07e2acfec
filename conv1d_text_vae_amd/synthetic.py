"""Synthetic workload of the named shapes (SURVEY.md 8d / BASELINE.md): no dataset or FastText download is possible
offline, so benchmarks and smoke tests use seeded random embeddings laid out exactly as the reference lays out real
ones (prepare_vocabulary_and_word_vectors :1051-1061, SequenceForVAE.__getitem__ :1374-1403)."""
from __future__ import annotations

import numpy as np


def synthetic_vocab(n_words: int, emb: int, seed: int = 1234) -> np.ndarray:
    """[n_words + 2, emb + 2] float32: unit-norm rows, then the unknown-word row (slot emb) and the EOS row (last slot)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    W = np.zeros((n_words + 2, emb + 2), dtype=np.float32)
    v = rng.standard_normal((n_words, emb), dtype=np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    W[:n_words, :emb] = v
    W[n_words, emb] = 1.0
    W[n_words + 1, emb + 1] = 1.0
    return W


def synthetic_batch(W_in: np.ndarray, n_out_rows: int, batch: int, t_in: int, t_out: int, seed: int = 1234,
                    with_input_ids: bool = False):
    """x [B, T_in, V] float32 and target ids [B, T_out] int32 (-1 = no target): word rows for t < len, the EOS row at
    t == len, zeros after; lengths ~ U{3..T-1}.  with_input_ids: also the rows of W_in that make up x, [B, T_in] int32
    (-1 = all-zero row) -- the form fit() feeds the engine (c1v_gather_input)."""
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    n_in = W_in.shape[0]
    in_ids = np.full((batch, t_in), -1, dtype=np.int32)
    ids = np.full((batch, t_out), -1, dtype=np.int32)
    li = rng.integers(3, t_in, size=batch)
    lo = rng.integers(3, t_out, size=batch)
    for b in range(batch):
        in_ids[b, :li[b]] = rng.integers(0, n_in - 2, size=li[b])
        in_ids[b, li[b]] = n_in - 1
        ids[b, :lo[b]] = rng.integers(0, n_out_rows - 2, size=lo[b])
        ids[b, lo[b]] = n_out_rows - 1
    x = np.zeros((batch, t_in, W_in.shape[1]), dtype=np.float32)
    keep = in_ids >= 0
    x[keep] = W_in[in_ids[keep]]
    return (x, ids, in_ids) if with_input_ids else (x, ids)
