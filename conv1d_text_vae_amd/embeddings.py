"""Embedding contract of the estimator (reference: the return type of conv1d_text_vae/fasttext_loading.py:11-59).

The reference takes gensim `FastTextKeyedVectors` objects and touches exactly these members:
  * `.vector_size`                                  (conv1d_text_vae.py:815)
  * `emb[word] -> np.ndarray[vector_size]`, raising for a word without a vector   (:852-854, :1021-1023)
  * `.similar_by_vector(vector, topn)` -> [(word, cosine similarity)], best first (:587)
  * `.save(path)` / `FastTextKeyedVectors.load(path)`                            (:668-669)
gensim is not installable offline, so this module ships a dependency-free container with those members (and the FastText
sub-word fallback: an out-of-vocabulary word gets the mean of its hashed character n-gram vectors).  When gensim IS
importable its own class is accepted as well (`KEYED_VECTOR_TYPES`).

`similar_by_vector` is row a10-ii of the hot path: it runs on the GPU through `c1v_vocab_topn`
(fp32 MFMA GEMM + fused row top-n).  There is no CPU implementation here.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

try:                                                     # pragma: no cover - gensim is absent in this image
    from gensim.models.keyedvectors import FastTextKeyedVectors as _GensimFastTextKeyedVectors
except Exception:                                        # noqa: BLE001
    _GensimFastTextKeyedVectors = None


def ft_hash(ngram: str) -> int:
    """FastText's FNV-1a variant over the UTF-8 bytes (each byte sign-extended, as the C++ `int8_t` cast does)."""
    h = 2166136261
    for b in ngram.encode('utf-8'):
        if b >= 128:
            b -= 256
        h = (h ^ (b & 0xFFFFFFFF)) & 0xFFFFFFFF
        h = (h * 16777619) & 0xFFFFFFFF
    return h


def compute_ngrams(word: str, min_n: int, max_n: int) -> List[str]:
    """Character n-grams of `<word>` with min_n <= n <= max_n (FastText's sub-word units)."""
    ext = '<' + word + '>'
    out = []
    for n in range(min_n, min(len(ext), max_n) + 1):
        for i in range(0, len(ext) - n + 1):
            out.append(ext[i:i + n])
    return out


class FastTextKeyedVectors:
    """Keyed word vectors with FastText sub-word fallback; the members the reference uses, nothing else."""

    def __init__(self, vector_size: int, min_n: int, max_n: int, bucket: int = 0):
        self.vector_size = int(vector_size)
        self.min_n = int(min_n)
        self.max_n = int(max_n)
        self.bucket = int(bucket)
        self.index2word: List[str] = []
        self.vocab: Dict[str, int] = {}
        self.vectors = np.zeros((0, self.vector_size), dtype=np.float32)
        self.vectors_ngrams = np.zeros((self.bucket, self.vector_size), dtype=np.float32)
        self._topn = None                                   # lazily built device index (engine.VocabTopN)
        # Opt-in, for tables too large for one GPU or a decode spread over the ranks of a torchrun job: every rank keeps
        # rows [begin, end) of the table on its GPU and `similar_by_vectors` becomes a COLLECTIVE (dist.ShardedVocabTopN:
        # every rank must call it with the same queries; per-shard lists are all-gathered and merged).  Off by default --
        # a single process, or ranks that decode different queries, need the whole table each.
        self.shard_over_ranks = False

    # ---- construction ----------------------------------------------------------------------------------
    def add(self, words: Sequence[str], vectors: np.ndarray):
        vectors = np.asarray(vectors, dtype=np.float32)
        if vectors.ndim != 2 or vectors.shape[1] != self.vector_size or vectors.shape[0] != len(words):
            raise ValueError('vectors must be [{0}, {1}]'.format(len(words), self.vector_size))
        base = len(self.index2word)
        for i, w in enumerate(words):
            if w in self.vocab:
                raise ValueError('word `{0}` is already in the vocabulary'.format(w))
            self.vocab[w] = base + i
            self.index2word.append(w)
        self.vectors = np.concatenate([self.vectors, vectors], axis=0)
        self._topn = None

    # ---- lookup ----------------------------------------------------------------------------------------
    def __contains__(self, word: str) -> bool:
        if word in self.vocab:
            return True
        return self.bucket > 0 and len(compute_ngrams(word, self.min_n, self.max_n)) > 0

    def __len__(self) -> int:
        return len(self.index2word)

    def word_vec(self, word: str) -> np.ndarray:
        idx = self.vocab.get(word)
        if idx is not None:
            return self.vectors[idx]
        if self.bucket <= 0:
            raise KeyError('word `{0}` not in vocabulary'.format(word))
        ngrams = compute_ngrams(word, self.min_n, self.max_n)
        if len(ngrams) == 0:
            raise KeyError('all ngrams for word `{0}` absent from model'.format(word))
        rows = [ft_hash(g) % self.bucket for g in ngrams]
        return self.vectors_ngrams[rows].sum(axis=0) / np.float32(len(rows))

    def __getitem__(self, word):
        if isinstance(word, str):
            return self.word_vec(word)
        return np.vstack([self.word_vec(w) for w in word])

    @property
    def vectors_norm(self) -> np.ndarray:
        n = np.linalg.norm(self.vectors, axis=1, keepdims=True)
        n[n == 0.0] = 1.0
        return (self.vectors / n).astype(np.float32)

    def similar_by_vector(self, vector: np.ndarray, topn: int = 10, device: str = 'cuda:0') -> List[Tuple[str, float]]:
        """gensim semantics: cosine similarity of `vector` against every vocabulary word, the `topn` best first
        (ties keep vocabulary order).  Runs on the GPU (c1v_vocab_topn); fails without one."""
        res = self.similar_by_vectors(np.asarray(vector, dtype=np.float32)[None, :], topn, device)
        return res[0]

    def similar_by_vectors(self, vectors: np.ndarray, topn: int = 10, device: str = 'cuda:0'):
        """Batched `similar_by_vector`: vectors [M, vector_size] -> M lists of (word, similarity)."""
        from .engine import VocabTopN
        q = np.asarray(vectors, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.vector_size:
            raise ValueError('vectors must be [M, {0}]'.format(self.vector_size))
        if len(self.index2word) == 0:
            return [[] for _ in range(q.shape[0])]
        norms = np.linalg.norm(q, axis=1, keepdims=True)
        norms[norms == 0.0] = 1.0
        topn = int(min(topn, len(self.index2word)))
        if self._topn is None or self._topn.device_str != device:
            if self.shard_over_ranks:
                from .dist import ShardedVocabTopN
                self._topn = ShardedVocabTopN(self.vectors_norm, device=device)
            else:
                self._topn = VocabTopN(self.vectors_norm, device=device)
        score, idx = self._topn.run(q / norms, topn)
        return [[(self.index2word[int(idx[m, j])], float(score[m, j])) for j in range(topn)] for m in range(q.shape[0])]

    # ---- persistence (reference :668-669; `remove_fasttext_files` removes `<name>` and `<name>*.npy`) ----
    def save(self, fname: str):
        with open(fname, 'wb') as fp:
            np.savez(fp, vector_size=self.vector_size, min_n=self.min_n, max_n=self.max_n, bucket=self.bucket,
                     words=np.array(self.index2word, dtype=object).astype(str) if self.index2word else np.array([], dtype=str),
                     vectors=self.vectors, vectors_ngrams=self.vectors_ngrams)

    @classmethod
    def load(cls, fname: str) -> 'FastTextKeyedVectors':
        with np.load(fname, allow_pickle=False) as z:
            res = cls(int(z['vector_size']), int(z['min_n']), int(z['max_n']), int(z['bucket']))
            words = [str(w) for w in z['words']]
            res.add(words, z['vectors'])
            res.vectors_ngrams = z['vectors_ngrams'].astype(np.float32)
        return res

    def __getstate__(self):
        st = dict(self.__dict__)
        st['_topn'] = None                                  # device index is rebuilt on demand
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self.__dict__.setdefault('shard_over_ranks', False)


KEYED_VECTOR_TYPES = (FastTextKeyedVectors,) if _GensimFastTextKeyedVectors is None else \
    (FastTextKeyedVectors, _GensimFastTextKeyedVectors)


def expected_type():
    """The type named in `check_embeddings`' error text (reference :573-574 builds a throw-away instance for it)."""
    return type((_GensimFastTextKeyedVectors or FastTextKeyedVectors)(vector_size=300, min_n=1, max_n=5))


def synthetic_fasttext(words: Iterable[str], vector_size: int = 300, min_n: int = 3, max_n: int = 6,
                       bucket: int = 20000, seed: int = 1234) -> FastTextKeyedVectors:
    """Deterministic stand-in for a downloaded FastText model (no network here): N(0,1) sub-word vectors in `bucket`
    hash rows; every listed word gets the mean of its n-gram vectors as its in-vocabulary vector, so related word forms
    are close, and out-of-vocabulary words still get vectors, as with real FastText."""
    rng = np.random.Generator(np.random.PCG64(seed))
    kv = FastTextKeyedVectors(vector_size, min_n, max_n, bucket)
    kv.vectors_ngrams = rng.standard_normal((bucket, vector_size)).astype(np.float32)
    uniq, seen = [], set()
    for w in words:
        if w not in seen and len(w) > 0:
            seen.add(w)
            uniq.append(w)
    if uniq:
        kv.add(uniq, np.vstack([kv.word_vec(w) for w in uniq]))
    return kv


def load_word2vec_text(path: str, limit: Optional[int] = None) -> FastTextKeyedVectors:
    """Reader for the textual `.vec` format FastText publishes next to its `.bin` files (first line `n dim`, then one
    word and `dim` floats per line), given a LOCAL path: real embeddings without gensim and without network."""
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    words, rows = [], []
    with open(path, 'r', encoding='utf-8', errors='ignore') as fp:
        header = fp.readline().split()
        n, dim = int(header[0]), int(header[1])
        for line in fp:
            parts = line.rstrip('\n').split(' ')
            if len(parts) < dim + 1:
                continue
            words.append(parts[0])
            rows.append(np.asarray(parts[1:dim + 1], dtype=np.float32))
            if limit is not None and len(words) >= limit:
                break
    kv = FastTextKeyedVectors(dim, 0, 0, 0)
    kv.add(words, np.vstack(rows) if rows else np.zeros((0, dim), dtype=np.float32))
    return kv


FASTTEXT_MAGIC = 793712314          # FASTTEXT_FILEFORMAT_MAGIC_INT32 of the facebookresearch/fastText binary format
FASTTEXT_VERSION = 12


def load_fasttext_bin(path: str, limit: Optional[int] = None) -> FastTextKeyedVectors:
    """Reader for a native FastText `.bin` model (`cc.ru.300.bin`, `cc.en.300.bin`, ...; a `.gz` of one is opened in
    place) given a LOCAL path.  The reference obtains the same object by downloading such a file and handing it to gensim
    (fasttext_loading.py:27, :43, :59); gensim and the network are absent here, so the published layout of the file is
    read directly:

        int32 magic, int32 version
        args:  12 x int32 (dim, ws, epoch, minCount, neg, wordNgrams, loss, model, bucket, minn, maxn, lrUpdateRate), double t
        dictionary:  int32 size, int32 nwords, int32 nlabels, int64 ntokens, int64 pruneidx_size,
                     size x (zero-terminated UTF-8 word, int64 count, int8 type), pruneidx_size x (int32, int32)
        input matrix:  bool quantised, int64 rows (= nwords + bucket), int64 dim, rows x dim float32
        (output matrix: not needed for word vectors, not read)

    Vectors follow gensim's `FastTextKeyedVectors`: an in-vocabulary word is the mean of its own input row and the rows of
    its character n-grams (hashed into `bucket` rows behind the words); an out-of-vocabulary word is the mean of its
    n-gram rows.  `limit` keeps only the first words of the dictionary (they are sorted by frequency).  Quantised
    (`.ftz`) models and pruned dictionaries are refused."""
    import gzip
    import struct
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    opener = gzip.open if path.endswith('.gz') else open
    with opener(path, 'rb') as fp:
        def take(fmt):
            size = struct.calcsize(fmt)
            raw = fp.read(size)
            if len(raw) != size:
                raise ValueError('`{0}` is not a FastText model: the file ends inside its header.'.format(path))
            return struct.unpack(fmt, raw)
        magic, version = take('<ii')
        if magic != FASTTEXT_MAGIC:
            raise ValueError('`{0}` is not a FastText model (magic number {1}).'.format(path, magic))
        if version > FASTTEXT_VERSION:
            raise ValueError('`{0}`: unsupported FastText format version {1}.'.format(path, version))
        dim, _ws, _epoch, _min_count, _neg, _word_ngrams, _loss, _model, bucket, minn, maxn, _lr_update = take('<12i')
        take('<d')
        size, nwords, _nlabels = take('<3i')
        _ntokens, pruneidx_size = take('<qq')
        if pruneidx_size > 0:
            raise ValueError('`{0}`: pruned (quantised) FastText dictionaries are not supported.'.format(path))
        words: List[str] = []
        for _ in range(size):
            chunk = bytearray()
            while True:
                ch = fp.read(1)
                if not ch:
                    raise ValueError('`{0}`: the dictionary is truncated.'.format(path))
                if ch == b'\x00':
                    break
                chunk += ch
            _count, kind = take('<qb')
            if kind == 0:                                   # 0 = word, 1 = label of a supervised model
                words.append(chunk.decode('utf-8', errors='ignore'))
        if len(words) != nwords:
            raise ValueError('`{0}`: the dictionary holds {1} words, its header says {2}.'.format(path, len(words), nwords))
        quantised, = take('<?')
        if quantised:
            raise ValueError('`{0}`: quantised FastText models are not supported.'.format(path))
        rows, cols = take('<qq')
        if cols != dim or rows != nwords + bucket:
            raise ValueError('`{0}`: the input matrix is {1} x {2}, expected {3} x {4}.'.format(
                path, rows, cols, nwords + bucket, dim))

        def read_rows(n):
            want = n * dim * 4
            raw = fp.read(want)
            if len(raw) != want:
                raise ValueError('`{0}`: the input matrix is truncated.'.format(path))
            return np.frombuffer(raw, dtype='<f4').reshape(n, dim).astype(np.float32)
        keep = nwords if limit is None else max(0, min(int(limit), nwords))
        vocab_rows = read_rows(keep)
        if keep < nwords:                                    # skip the rows of the words beyond `limit`
            skip = (nwords - keep) * dim * 4
            while skip > 0:
                got = fp.read(min(skip, 1 << 24))
                if not got:
                    raise ValueError('`{0}`: the input matrix is truncated.'.format(path))
                skip -= len(got)
        ngram_rows = read_rows(bucket)
    kv = FastTextKeyedVectors(dim, minn, maxn, bucket)
    kv.vectors_ngrams = ngram_rows
    final = np.empty_like(vocab_rows)
    for i, w in enumerate(words[:keep]):
        ngrams = compute_ngrams(w, minn, maxn) if bucket > 0 else []
        acc = vocab_rows[i].astype(np.float32).copy()
        if ngrams:
            acc += ngram_rows[[ft_hash(g) % bucket for g in ngrams]].sum(axis=0)
        final[i] = acc / np.float32(len(ngrams) + 1)
    kv.add(words[:keep], final)
    return kv
