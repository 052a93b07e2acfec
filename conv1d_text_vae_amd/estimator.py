"""`Conv1dTextVAE`: the reference's estimator surface (conv1d_text_vae/conv1d_text_vae.py:141-1343) over the MI355X
hot path.

Drop-in contract: constructor hyper-parameters, `fit / transform / predict / fit_transform /
fit_predict / get_params / set_params`, the static helpers the reference's tests call, the exact `ValueError` texts,
the fitted attributes and pickling that drops the embeddings.  What differs underneath:

  * the four Keras models are facades (`predict_on_batch`, NumPy in / out) over ONE `VAEEngine` (hand-written HIP
    kernels behind libc1dvae.so) and one `Seq2SeqGenerator` (stock PyTorch-ROCm);
  * `fit` never builds the dense one-hot target `[B, T, N]` (:1372): batches are int32 word ids, the encoder operand is
    gathered on the device (`c1v_gather_input`), and a whole train step is one captured HIP graph;
  * weights are pickled as `.npz` bytes keyed by the Keras layer names, in Keras layouts, instead of HDF5 bytes.

There is no CPU fallback: everything numerical raises `C1VError` when libc1dvae.so or the GPU is missing.
"""
from __future__ import annotations

import copy
import io
import time
import math
import os
import tempfile
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
from scipy.spatial import distance
from sklearn.base import BaseEstimator, ClassifierMixin, TransformerMixin
from sklearn.utils.validation import check_is_fitted

from . import embeddings as E
from .geometry import Geometry, init_keras_weights
from .tokenizer import BaseTokenizer, DefaultTokenizer
from .training import run_epochs, run_epochs_dp

EPSILON = 1e-7                    # keras.backend.epsilon()
FITTED_ATTRIBUTES = ['input_text_size_', 'output_text_size_', 'input_vector_size_', 'output_vector_size_',
                     'vae_encoder_', 'vae_decoder_', 'generator_encoder_', 'generator_decoder_',
                     'output_text_size_in_characters_', 'target_char_index_', 'reverse_target_char_index_']
# layers of 'EncoderForVAE' = Model(encoder_input, z_mean) (:1308): what `vae_encoder_` saves, pickles and hands to a warm start.
# The z_log_var head belongs to neither saved model: after unpickling / in a warm start it is freshly initialised, as in the
# reference (whose rebuilt graph gives it new random weights, frozen in a warm start, :1245).
ENCODER_PREFIXES = ('encoder_conv1d_', 'encoder_batchnorm_', 'z_mean')
UNSAVED_PREFIXES = ('z_log_var',)


def _npz_bytes(weights: Dict[str, np.ndarray]) -> bytes:
    buf = io.BytesIO()
    np.savez(buf, **{k.replace('/', '__'): v for k, v in weights.items()})
    return buf.getvalue()


def _npz_load(data: Union[bytes, bytearray]) -> Dict[str, np.ndarray]:
    if (not isinstance(data, bytearray)) and (not isinstance(data, bytes)):
        raise ValueError('The `weights_as_bytes` must be an array of bytes, not `{0}`!'.format(type(data)))
    with np.load(io.BytesIO(bytes(data)), allow_pickle=False) as z:
        return {k.replace('__', '/'): z[k] for k in z.files}


# ------------------------------------------------------------------------------------------------------------
# runtime shared by the model facades
# ------------------------------------------------------------------------------------------------------------
class _VAERuntime:
    """Geometry + weights + the lazily created `VAEEngine` (creating it needs the GPU; a pickled estimator can be loaded
    anywhere and only fails, loudly, when it is asked to compute)."""

    def __init__(self, geom: Geometry, batch: int, dtype: str, device: str, weights: Dict[str, np.ndarray], seed: int = 0):
        self.geom, self.batch, self.dtype, self.device, self.seed = geom, int(batch), dtype, device, int(seed)
        self._weights = weights
        self._engine = None

    def engine(self):
        """The engine for inference; created on first use (no vocabulary matrix: encode / decode only)."""
        if self._engine is None:
            self._engine = self._make(None, False)
        return self._engine

    def training_engine(self, vocab_vectors: np.ndarray, frozen_encoder: bool = False, dp=None):
        """A fresh engine that also owns the vocabulary matrix W of the loss and the optimizer state.  dp (dist.DataParallel,
        world > 1): this rank's engine of a data-parallel fit -- `batch / world` samples per step, gradients all-reduced."""
        w = self.weights()
        self._engine = None
        if dp is not None and dp.world > 1:
            _, local = dp.batch_slice(self.batch)
            self._engine = self._make(vocab_vectors, frozen_encoder, w, batch=local, world=dp.world, rank=dp.rank)
            self._engine_is_rank_slice = True
        else:
            self._engine = self._make(vocab_vectors, frozen_encoder, w)
        return self._engine

    def _make(self, vocab_vectors, frozen_encoder, weights=None, batch=None, world=1, rank=0):
        from .engine import VAEEngine
        return VAEEngine(self.geom, self.batch if batch is None else batch, vocab_vectors, dtype=self.dtype,
                         device=self.device, seed=self.seed, world_size=world, rank=rank, frozen_encoder=frozen_encoder,
                         weights=weights if weights is not None else self._weights)

    def weights(self) -> Dict[str, np.ndarray]:
        if self._engine is not None:
            self._weights = self._engine.get_weights()
        return self._weights

    def release(self, keep_weights: bool = True):
        if self._engine is not None and keep_weights:
            self._weights = self._engine.get_weights()
        self._engine = None

    def _batched(self, x: np.ndarray, fn):
        """Run `fn` (which consumes the engine's resident batch) over an array with any number of rows."""
        B = self.batch
        outs = []
        for s in range(0, x.shape[0], B):
            part = x[s:s + B]
            n = part.shape[0]
            if n < B:
                part = np.concatenate([part, np.zeros((B - n,) + part.shape[1:], dtype=part.dtype)], axis=0)
            outs.append(fn(part)[:n])
        return outs[0] if len(outs) == 1 else np.concatenate(outs, axis=0)


class VAEEncoderModel:
    """`vae_encoder_` ('EncoderForVAE', :1308): predict_on_batch(x [B, T_in, V_in]) -> z_mean [B, latent]."""

    def __init__(self, rt: _VAERuntime):
        self.rt = rt

    def predict_on_batch(self, x) -> np.ndarray:
        x = np.asarray(x, dtype=np.float32)
        g = self.rt.geom
        if x.ndim != 3 or x.shape[1] != g.input_text_size or x.shape[2] != g.input_vector_size:
            raise ValueError('Expected an array of shape (batch, {0}, {1}), got {2}.'.format(
                g.input_text_size, g.input_vector_size, x.shape))
        eng = self.rt.engine()

        def run(part):
            eng.load_batch(x=part)
            return eng.encode().cpu().numpy()
        return self.rt._batched(x, run)

    predict = predict_on_batch

    def get_weights(self) -> Dict[str, np.ndarray]:
        return {k: v for k, v in self.rt.weights().items() if k.startswith(ENCODER_PREFIXES)}


class VAEDecoderModel:
    """`vae_decoder_` ('DecoderForVAE', :1281): predict_on_batch(z [B, latent]) -> y [B, T_out, V_out]."""

    def __init__(self, rt: _VAERuntime):
        self.rt = rt

    def predict_on_batch(self, z) -> np.ndarray:
        import torch
        z = np.asarray(z, dtype=np.float32)
        if z.ndim != 2 or z.shape[1] != self.rt.geom.latent_dim:
            raise ValueError('Expected an array of shape (batch, {0}), got {1}.'.format(self.rt.geom.latent_dim, z.shape))
        eng = self.rt.engine()

        def run(part):
            return eng.decode_latent(torch.from_numpy(part).to(eng.dev)).cpu().numpy()
        return self.rt._batched(z, run)

    predict = predict_on_batch

    def get_weights(self) -> Dict[str, np.ndarray]:
        return {k: v for k, v in self.rt.weights().items() if not k.startswith(ENCODER_PREFIXES + UNSAVED_PREFIXES)}


class FullVAEModel:
    """`vae_model_for_training` used as a predictor ('FullVAE', :1320, :1462-1464): dropout off, z SAMPLED."""

    def __init__(self, rt: _VAERuntime):
        self.rt = rt

    def predict_on_batch(self, x) -> np.ndarray:
        x = np.asarray(x, dtype=np.float32)
        eng = self.rt.engine()

        def run(part):
            eng.load_batch(x=part)
            return eng.sample_reconstruct(postprocess=False).cpu().numpy().copy()
        return self.rt._batched(x, run)

    predict = predict_on_batch


# ------------------------------------------------------------------------------------------------------------
# batch generators with the reference's interface
# ------------------------------------------------------------------------------------------------------------
def encode_word_ids(texts, vocabulary: dict, text_size: int, zero_after_eos: bool = True) -> np.ndarray:
    """int32 [n, text_size]: the row of `vocabulary` per step; the step right after the last token carries the
    end-of-sentence row `vocabulary['']`; later steps carry -1 (an all-zero vector / no target) when `zero_after_eos`
    (SequenceForVAE :1380-1403) or the end-of-sentence row again (SequenceForSeq2Seq :1444-1449)."""
    eos = vocabulary['']
    ids = np.full((len(texts), text_size), -1 if zero_after_eos else eos, dtype=np.int32)
    for i, text in enumerate(texts):
        n = min(len(text), text_size)
        for t in range(n):
            ids[i, t] = vocabulary[text[t]]
        if n < text_size:
            ids[i, n] = eos
    return ids


class SequenceForVAE:
    """Reference :1346-1405.  `__getitem__` keeps the reference contract (dense float32 input, dense one-hot target);
    the estimator itself trains from `ids(idx)` (int32 word ids), which describes the same batch in 1/300th of the bytes."""

    def __init__(self, tokenizer: BaseTokenizer, input_texts: tuple, target_texts: tuple, batch_size: int,
                 input_text_size: int, output_text_size: int, input_vocabulary: dict, input_word_vectors: np.ndarray,
                 output_vocabulary: dict, output_word_vectors: np.ndarray):
        self.tokenizer = tokenizer
        self.input_texts = input_texts
        self.target_texts = target_texts
        self.batch_size = batch_size
        self.input_text_size = input_text_size
        self.output_text_size = output_text_size
        self.n_text_pairs = len(input_texts)
        self.n_batches = self.n_text_pairs // self.batch_size
        self.input_vocabulary = input_vocabulary
        self.output_vocabulary = output_vocabulary
        self.input_word_vectors = input_word_vectors
        self.output_word_vectors = output_word_vectors
        self._in_ids = None
        self._out_ids = None

    def __len__(self):
        return self.n_batches

    def all_ids(self) -> Tuple[np.ndarray, np.ndarray]:
        if self._in_ids is None:
            self._in_ids = encode_word_ids(self.input_texts, self.input_vocabulary, self.input_text_size)
            self._out_ids = encode_word_ids(self.target_texts, self.output_vocabulary, self.output_text_size)
        return self._in_ids, self._out_ids

    def _rows(self, idx: int) -> np.ndarray:
        return np.arange(idx * self.batch_size, (idx + 1) * self.batch_size) % max(self.n_text_pairs, 1)

    def ids(self, idx: int) -> Tuple[np.ndarray, np.ndarray]:
        a, b = self.all_ids()
        r = self._rows(idx)
        return a[r], b[r]

    def __getitem__(self, idx):
        in_ids, out_ids = self.ids(idx)
        input_data = np.zeros((self.batch_size, self.input_text_size, self.input_word_vectors.shape[1]), dtype=np.float32)
        keep = in_ids >= 0
        input_data[keep] = self.input_word_vectors[in_ids[keep]]
        target_data = np.zeros((self.batch_size, self.output_text_size, self.output_word_vectors.shape[0]), dtype=np.float32)
        b, t = np.nonzero(out_ids >= 0)
        target_data[b, t, out_ids[b, t]] = 1.0
        return input_data, target_data


class SequenceForSeq2Seq:
    """Reference :1408-1465: ([postprocessed VAE reconstruction, teacher-forcing characters], next characters)."""

    def __init__(self, tokenizer: BaseTokenizer, input_texts: tuple, target_texts: tuple, batch_size: int,
                 input_text_size: int, output_text_size: int, input_vocabulary: dict, input_word_vectors: np.ndarray,
                 output_char_index: dict, vae):
        self.tokenizer = tokenizer
        self.input_texts = input_texts
        self.target_texts = target_texts
        self.batch_size = batch_size
        self.input_text_size = input_text_size
        self.output_text_size = output_text_size
        self.n_text_pairs = len(input_texts)
        self.n_batches = self.n_text_pairs // self.batch_size
        self.input_vocabulary = input_vocabulary
        self.input_word_vectors = input_word_vectors
        self.output_char_index = output_char_index
        self.vae = vae
        self._in_ids = None
        self._char_in = None
        self._char_out = None

    def __len__(self):
        return self.n_batches

    def all_ids(self):
        """word ids [n, T_in] (EOS row repeated after the text) and character ids [n, L] for the decoder input / target
        (-1 = all-zero one-hot)."""
        if self._in_ids is None:
            self._in_ids = encode_word_ids(self.input_texts, self.input_vocabulary, self.input_text_size, zero_after_eos=False)
            L = self.output_text_size
            bos = self.output_char_index[Conv1dTextVAE.SEQUENCE_BEGIN]
            eos = self.output_char_index[Conv1dTextVAE.SEQUENCE_END]
            cin = np.full((self.n_text_pairs, L), -1, dtype=np.int32)
            cout = np.full((self.n_text_pairs, L), -1, dtype=np.int32)
            for i, chars in enumerate(self.target_texts):
                T = min(len(chars), L - 3)
                cin[i, 0] = bos
                for t in range(T):
                    c = self.output_char_index[chars[t]]
                    cin[i, t + 1] = c
                    cout[i, t] = c
                cin[i, T + 1] = eos
                cout[i, T] = eos
                cout[i, T + 1] = eos
            self._char_in, self._char_out = cin, cout
        return self._in_ids, self._char_in, self._char_out

    def _rows(self, idx: int) -> np.ndarray:
        return np.arange(idx * self.batch_size, (idx + 1) * self.batch_size) % max(self.n_text_pairs, 1)

    def __getitem__(self, idx):
        in_ids, cin, cout = self.all_ids()
        r = self._rows(idx)
        input_data = self.input_word_vectors[in_ids[r]].astype(np.float32)
        n_chars = len(self.output_char_index)
        gen_in = np.zeros((self.batch_size, self.output_text_size, n_chars), dtype=np.float32)
        gen_out = np.zeros((self.batch_size, self.output_text_size, n_chars), dtype=np.float32)
        b, t = np.nonzero(cin[r] >= 0)
        gen_in[b, t, cin[r][b, t]] = 1.0
        b, t = np.nonzero(cout[r] >= 0)
        gen_out[b, t, cout[r][b, t]] = 1.0
        reconstructed = Conv1dTextVAE.postprocess_reconstructed_word_vectors(self.vae.predict_on_batch(input_data))
        return [reconstructed, gen_in], gen_out


# ------------------------------------------------------------------------------------------------------------
# the estimator
# ------------------------------------------------------------------------------------------------------------
class Conv1dTextVAE(BaseEstimator, TransformerMixin, ClassifierMixin):
    SEQUENCE_BEGIN = '<BOS>'
    SEQUENCE_END = '<EOS>'
    # not hyper-parameters of the reference, hence class attributes and not constructor arguments (get_params stays
    # identical): arithmetic type of the HIP path ('bf16' = bf16 MFMA with fp32 accumulation and fp32 master weights,
    # 'f32' = fp32 MFMA) and the device it runs on
    compute_dtype = os.environ.get('C1V_DTYPE', 'bf16')
    device = 'cuda:0'

    def __init__(self, input_embeddings, output_embeddings, tokenizer: BaseTokenizer = None,
                 n_filters: Union[int, tuple] = 128, kernel_size: int = 3, latent_dim: int = 5,
                 n_recurrent_units: int = 128, input_text_size: int = None, output_text_size: int = None,
                 batch_size: int = 64, max_epochs: int = 100, lr: float = 0.001, validation_fraction: float = 0.2,
                 use_batch_norm: bool = False, use_attention: bool = True, output_onehot_size: int = None,
                 warm_start: bool = False, verbose: bool = False):
        self.n_filters = n_filters
        self.kernel_size = kernel_size
        self.input_embeddings = input_embeddings
        self.output_embeddings = output_embeddings
        self.batch_size = batch_size
        self.max_epochs = max_epochs
        self.lr = lr
        self.latent_dim = latent_dim
        self.warm_start = warm_start
        self.verbose = verbose
        self.input_text_size = input_text_size
        self.output_text_size = output_text_size
        self.validation_fraction = validation_fraction
        self.n_recurrent_units = n_recurrent_units
        self.tokenizer = tokenizer
        self.output_onehot_size = output_onehot_size
        self.use_batch_norm = use_batch_norm
        self.use_attention = use_attention

    def __del__(self):
        try:
            self._release_models(keep_weights=False)
        except Exception:                                               # noqa: BLE001 - interpreter shutdown
            pass

    def _release_models(self, keep_weights: bool = True):
        rt = self.__dict__.get('_vae_runtime')
        if rt is not None:
            rt.release(keep_weights)
        for name in ('vae_encoder_', 'vae_decoder_', 'generator_encoder_', 'generator_decoder_'):
            self.__dict__.pop(name, None)

    # ---- fit ---------------------------------------------------------------------------------------------
    def fit(self, X, y=None):
        self.check_params(**self.get_params(deep=False))
        self.check_embeddings()
        self.check_texts_param(X, 'X')
        y_ = X
        if y is not None:                                            # y = None: plain auto-encoding of X
            self.check_texts_param(y, 'y')
            if len(y) != len(X):
                raise ValueError('Length of `X` does not equal to length of `y`! {0} != {1}.'.format(len(X), len(y)))
            y_ = y
        if self.tokenizer is None:
            self.tokenizer = DefaultTokenizer()
        # the LAST round(n * validation_fraction) pairs are the evaluation set (:201-217); both parts must be non-empty
        n_eval_set = int(round(len(X) * self.validation_fraction))
        if not 1 <= n_eval_set < len(X):
            raise ValueError('`validation_fraction` is too small! There are no samples for evaluation!' if n_eval_set < 1 else
                             '`validation_fraction` is too large! There are no samples for training!')
        special_symbols = self._special_symbols(none_if_empty=True)
        if self.warm_start:
            self.check_is_fitted()
            if self.input_vector_size_ != self.calc_vector_size(self.input_embeddings, special_symbols):
                raise ValueError('Input embeddings do not correspond to the neural network!')
        X, y_ = list(X), list(y_)
        n_train = len(X) - n_eval_set
        word_bounds_y = [self.tokenizer.tokenize_into_words(t) for t in y_]
        max_text_size = max(len(b) for b in word_bounds_y)
        if max_text_size == 0:
            raise ValueError('The parameters `y` is wrong! All texts are empty!')
        self.output_text_size_ = max_text_size if self.output_text_size is None else self.output_text_size
        input_texts = tuple(self.tokenize(t, self.tokenizer.tokenize_into_words(t)) for t in X)
        input_vocabulary, input_word_vectors = self.prepare_vocabulary_and_word_vectors(
            input_texts, self.input_embeddings, special_symbols)
        self.input_vector_size_ = input_word_vectors.shape[1]
        target_texts = tuple(self.tokenize(t, b) for t, b in zip(y_, word_bounds_y))
        target_texts_by_characters = tuple(tuple(self.tokenizer.tokenize_into_characters(t, b))
                                           for t, b in zip(y_, word_bounds_y))
        target_characters = set()
        for cur in target_texts_by_characters:
            target_characters |= set(cur)
        self.output_text_size_in_characters_ = max(len(cur) for cur in target_texts_by_characters) + 3
        target_characters = sorted(list(target_characters | {self.SEQUENCE_BEGIN, self.SEQUENCE_END}))
        self.target_char_index_ = dict([(char, i) for i, char in enumerate(target_characters)])
        self.reverse_target_char_index_ = dict((i, char) for char, i in self.target_char_index_.items())
        output_vocabulary, output_word_vectors = self.prepare_vocabulary_and_word_vectors(
            target_texts, self.input_embeddings if self.output_embeddings is None else self.output_embeddings,
            special_symbols, self.output_onehot_size, verbose=self.verbose)
        self.output_vector_size_ = output_word_vectors.shape[1]
        encoder_weights = None
        if self.warm_start:
            encoder_weights = self.vae_encoder_.get_weights()
        else:
            if self.input_text_size is None:
                size = max(len(self.tokenizer.tokenize_into_words(t)) for t in X)
                if size == 0:
                    raise ValueError('The parameters `X` is wrong! All texts are empty!')
                self.input_text_size_ = size
            else:
                self.input_text_size_ = self.input_text_size
        self._release_models()
        # Data parallel under torchrun (WORLD_SIZE > 1: one process per GPU, RCCL): `batch_size` stays the GLOBAL batch of one
        # optimizer step and every rank computes its contiguous share of it (dist.DataParallel, training.run_epochs_dp).
        # Rank 0's seed initialises the weights, the shuffles and the generator on every rank.
        from .dist import DataParallel
        dp = DataParallel.from_env()
        if dp.world > 1:
            dp.batch_slice(self.batch_size)                              # (raises if the batch does not divide)
            self.device = dp.device_str
        seed = dp.broadcast_int(int(np.random.randint(0, 2 ** 31 - 1)))
        chatty = bool(self.verbose) and dp.rank == 0
        runtime, generator = self._create_model(self.input_vector_size_, self.output_vector_size_, seed,
                                                encoder_weights=encoder_weights)
        engine = runtime.training_engine(output_word_vectors, frozen_encoder=bool(self.warm_start), dp=dp)
        if chatty:
            self._print_summary(runtime.geom, generator)
        self._fit_vae(engine, input_texts, target_texts, n_train, input_vocabulary, input_word_vectors,
                      output_vocabulary, output_word_vectors, dp=dp, seed=seed)
        self._fit_seq2seq(runtime, engine, generator, input_texts, target_texts_by_characters, n_train, input_vocabulary,
                          input_word_vectors, dp=dp, seed=seed)
        runtime.weights()
        if dp.world > 1:
            runtime.release(keep_weights=True)          # inference runs whole batches: a full-batch engine is built on demand
        self._vae_runtime = runtime
        self._generator = generator
        self.vae_encoder_ = VAEEncoderModel(runtime)
        self.vae_decoder_ = VAEDecoderModel(runtime)
        self.generator_encoder_ = generator.encoder_model()
        self.generator_decoder_ = generator.decoder_model()
        return self

    def _fit_vae(self, engine, input_texts, target_texts, n_train, input_vocabulary, input_word_vectors,
                 output_vocabulary, output_word_vectors, dp=None, seed=None):
        import torch
        from .dist import DataParallel
        dp = dp if dp is not None else DataParallel()
        chatty = bool(self.verbose) and dp.rank == 0
        all_reduce = dp.all_reduce_async if dp.world > 1 else None
        B = self.batch_size
        mk = lambda a, b: SequenceForVAE(
            input_texts=input_texts[a:b], target_texts=target_texts[a:b], tokenizer=self.tokenizer, batch_size=B,
            input_text_size=self.input_text_size_, output_text_size=self.output_text_size_,
            input_vocabulary=input_vocabulary, output_vocabulary=output_vocabulary,
            input_word_vectors=input_word_vectors, output_word_vectors=output_word_vectors)
        train_set, eval_set = mk(0, n_train), mk(n_train, len(input_texts))
        dev = engine.dev
        engine.set_input_table(input_word_vectors)
        # the whole data set as word ids lives in HBM (8 bytes per token position); a batch is a contiguous slice
        tr_in, tr_out = (torch.from_numpy(a).to(dev) for a in train_set.all_ids())
        ev_in, ev_out = (torch.from_numpy(a).to(dev) for a in eval_set.all_ids())
        acc = torch.zeros(4, dtype=torch.float64, device=dev)            # train loss sum, n, val loss sum, n

        # (first, count): this rank's samples of a global batch; the engine's loss already carries 1/B_global, so the
        # per-rank losses ADD UP to the batch loss (run_epochs_dp sums them over the ranks once per epoch)
        def train_slice(first, count):
            engine.in_ids.copy_(tr_in[first:first + count])
            engine.ids_in.copy_(tr_out[first:first + count])
            engine.train_step(from_ids=True, all_reduce=all_reduce)
            acc[0] += engine.loss_out[0]
            acc[1] += 1

        def eval_slice(first, count):
            engine.in_ids.copy_(ev_in[first:first + count])
            engine.ids_in.copy_(ev_out[first:first + count])
            engine.eval_step(from_ids=True)
            acc[2] += engine.loss_out[0]
            acc[3] += 1

        def local_sums():
            a = acc.cpu().numpy().copy()
            acc.zero_()
            return a

        # ModelCheckpoint(save_weights_only=True) saves every weight of the model (:312-316, :331-332): the trainable
        # parameters AND the moving averages of the BatchNormalization layers
        bn_state = [st[k] for st in engine.bn.values() for k in ('moving_mean', 'moving_var')]

        def snapshot():
            return [t.clone() for t in [engine.params] + bn_state]

        def restore(w):
            for t, v in zip([engine.params] + bn_state, w):
                t.copy_(v)
            engine._pack_all()

        if chatty:
            print('')
            print('----------------------------------------')
            print('VAE training...')
            print('----------------------------------------')
            print('')
        self.vae_history_ = run_epochs_dp(dp, B, len(train_set), len(eval_set), self.max_epochs, train_slice, eval_slice,
                                          local_sums, engine.get_lr, engine.set_lr, snapshot, restore,
                                          verbose=self.verbose, seed=seed, what='VAE')

    def _fit_seq2seq(self, runtime, engine, generator, input_texts, target_texts_by_characters, n_train, input_vocabulary,
                     input_word_vectors, dp=None, seed=None):
        import torch
        from .dist import DataParallel
        dp = dp if dp is not None else DataParallel()
        chatty = bool(self.verbose) and dp.rank == 0
        first, count = dp.batch_slice(self.batch_size)
        B = self.batch_size
        vae = FullVAEModel(runtime)
        mk = lambda a, b: SequenceForSeq2Seq(
            input_texts=input_texts[a:b], target_texts=target_texts_by_characters[a:b], tokenizer=self.tokenizer,
            batch_size=B, input_text_size=self.input_text_size_, output_text_size=self.output_text_size_in_characters_,
            input_vocabulary=input_vocabulary, input_word_vectors=input_word_vectors, vae=vae,
            output_char_index=self.target_char_index_)
        train_set, eval_set = mk(0, n_train), mk(n_train, len(input_texts))
        dev = engine.dev
        n_chars = len(self.target_char_index_)
        tr = [torch.from_numpy(a).to(dev) for a in train_set.all_ids()]
        ev = [torch.from_numpy(a).to(dev) for a in eval_set.all_ids()]
        acc = torch.zeros(4, dtype=torch.float64, device=dev)

        def onehot(ids):
            oh = torch.nn.functional.one_hot(ids.clamp(min=0).long(), n_chars).to(torch.float32)
            return oh * (ids >= 0).unsqueeze(-1).to(torch.float32)

        # Data parallel: every rank reconstructs ITS samples of the batch with the VAE (the sampling noise is keyed by the
        # global sample index), the word vectors are all-gathered, and the small recurrent generator trains replicated on
        # the whole batch (same seed, same inputs => identical weights on every rank, no gradient exchange).
        def batch(data, i):
            engine.in_ids.copy_(data[0][i * B + first:i * B + first + count])
            words = engine.sample_reconstruct(from_ids=True, postprocess=True)       # FullVAE.predict_on_batch + postprocess
            words = dp.all_gather_rows(torch.nan_to_num(words.clone(), nan=0.0, posinf=0.0, neginf=0.0))
            return words, onehot(data[1][i * B:(i + 1) * B]), onehot(data[2][i * B:(i + 1) * B])

        def train_batch(i):
            acc[0] += generator.train_batch(*batch(tr, i)).to(torch.float64)
            acc[1] += 1

        def eval_batch(i):
            acc[2] += generator.eval_batch(*batch(ev, i)).to(torch.float64)
            acc[3] += 1

        def epoch_losses():
            a = acc.cpu().numpy()
            acc.zero_()
            return float(a[0] / max(a[1], 1.0)), float(a[2] / max(a[3], 1.0))

        def get_lr():
            return generator.lr

        def set_lr(v):
            generator.lr = float(v)

        if chatty:
            print('')
            print('----------------------------------------')
            print('Seq2Seq training...')
            print('----------------------------------------')
            print('')
        shuffle = np.random.RandomState(None if seed is None else (int(seed) + 2) % (2 ** 31 - 1)) if dp.world > 1 else None
        self.seq2seq_history_ = run_epochs(len(train_set), len(eval_set), self.max_epochs, train_batch, eval_batch,
                                           epoch_losses, get_lr, set_lr, generator.get_weights, generator.set_weights,
                                           verbose=chatty, rng=shuffle, what='Seq2Seq')

    # ---- inference ---------------------------------------------------------------------------------------
    def _special_symbols(self, none_if_empty: bool = False):
        if hasattr(self.tokenizer, 'special_symbols') and self.tokenizer.special_symbols is not None:
            if none_if_empty and len(self.tokenizer.special_symbols) == 0:
                return None
            return tuple(sorted(list(self.tokenizer.special_symbols)))
        return None

    def transform(self, X) -> np.ndarray:
        self.check_is_fitted()
        self.check_embeddings()
        self.check_texts_param(X, 'X')
        if self.tokenizer is None:
            self.tokenizer = DefaultTokenizer()
        special_symbols = self._special_symbols()
        outputs = []
        n_done = 0
        for data_for_batch in self.texts_to_data(X, self.batch_size, self.input_text_size_, self.tokenizer,
                                                 self.input_embeddings, special_symbols):
            outputs_for_batch = self.vae_encoder_.predict_on_batch(data_for_batch)
            n = min(outputs_for_batch.shape[0], len(X) - n_done)
            outputs.append(outputs_for_batch[:n].copy())
            n_done += n
        return np.vstack(outputs) if len(outputs) > 0 else None

    def predict(self, X):
        import torch
        self.check_is_fitted()
        self.check_embeddings()
        self.check_texts_param(X, 'X')
        if self.tokenizer is None:
            self.tokenizer = DefaultTokenizer()
        special_symbols = self._special_symbols()
        chatty = isinstance(self.verbose, int) and (self.verbose > 1)
        generated_texts = []
        n_all_texts = len(X)
        n_data_parts = 20
        data_part_size = n_all_texts // n_data_parts
        data_part_counter = 0
        start_pos = 0
        if chatty:
            print('Prediction of texts with the VAE is started...')
        engine = self._vae_runtime.engine()
        gen = self._generator
        bos, eos = self.target_char_index_[self.SEQUENCE_BEGIN], self.target_char_index_[self.SEQUENCE_END]
        for data_for_batch in self.texts_to_data(X, self.batch_size, self.input_text_size_, self.tokenizer,
                                                 self.input_embeddings, special_symbols):
            # decoder(encoder(x)) -> EOS truncation (:439-441), all on the device; then the generator (:442-468)
            engine.load_batch(x=data_for_batch)
            words = engine.reconstruct(postprocess=True)
            words = torch.nan_to_num(words.clone(), nan=0.0, posinf=0.0, neginf=0.0).to(gen.dev)
            state = gen.encode(words, train=False)
            sampled = gen.greedy_decode(state, bos, eos, self.output_text_size_in_characters_)
            n_texts_in_batch = min(start_pos + data_for_batch.shape[0], n_all_texts) - start_pos
            for text_idx in range(n_texts_in_batch):
                row = sampled[text_idx]
                hits = np.nonzero(row == eos)[0]
                n = (int(hits[0]) + 1) if hits.size > 0 else row.shape[0]
                generated_texts.append(''.join(self.reverse_target_char_index_[int(c)] for c in row[:n]))
                if data_part_size > 0 and ((text_idx + start_pos + 1) % data_part_size) == 0:
                    data_part_counter += 1
                    if chatty:
                        print('{0}% of texts are processed...'.format(data_part_counter * (100 // n_data_parts)))
            start_pos += data_for_batch.shape[0]
        if chatty:
            if data_part_counter < n_data_parts:
                print('100% of texts are processed...')
            print('Prediction of texts with the VAE is finished...')
        return (np.array(generated_texts, dtype=object) if isinstance(X, np.ndarray) else (
            tuple(generated_texts) if isinstance(X, tuple) else generated_texts))

    def fit_transform(self, X, y=None, **fit_params):
        return self.fit(X, y).transform(X)

    def fit_predict(self, X, y=None, **fit_params):
        return self.fit(X, y).predict(X)

    # ---- parameters ------------------------------------------------------------------------------------------
    def get_params(self, deep=True):
        params = {
            'n_filters': copy.copy(self.n_filters) if deep else self.n_filters,
            'kernel_size': self.kernel_size,
            'batch_size': self.batch_size,
            'max_epochs': self.max_epochs,
            'lr': self.lr,
            'latent_dim': self.latent_dim,
            'n_recurrent_units': self.n_recurrent_units,
            'use_batch_norm': self.use_batch_norm,
            'use_attention': self.use_attention,
            'output_onehot_size': self.output_onehot_size,
            'warm_start': self.warm_start,
            'verbose': self.verbose,
            'input_text_size': self.input_text_size,
            'output_text_size': self.output_text_size,
            'validation_fraction': self.validation_fraction,
            'tokenizer': None if self.tokenizer is None else (copy.deepcopy(self.tokenizer) if deep else self.tokenizer),
        }
        if hasattr(self, 'input_embeddings'):
            params['input_embeddings'] = self.input_embeddings
        if hasattr(self, 'output_embeddings'):
            params['output_embeddings'] = self.output_embeddings
        return params

    def set_params(self, **params):
        self.n_filters = params['n_filters']
        self.kernel_size = params['kernel_size']
        self.input_embeddings = params.get('input_embeddings', None)
        self.output_embeddings = params.get('output_embeddings', None)
        for name in ('batch_size', 'max_epochs', 'lr', 'latent_dim', 'n_recurrent_units', 'use_batch_norm',
                     'use_attention', 'output_onehot_size', 'warm_start', 'verbose', 'input_text_size',
                     'output_text_size', 'validation_fraction', 'tokenizer'):
            setattr(self, name, params[name])
        return self

    def check_is_fitted(self):
        check_is_fitted(self, FITTED_ATTRIBUTES)

    def check_embeddings(self):
        if not hasattr(self, 'input_embeddings'):
            raise ValueError('The parameter `input_embeddings` is not defined!')
        if not hasattr(self, 'output_embeddings'):
            raise ValueError('The parameter `output_embeddings` is not defined!')
        if not isinstance(self.input_embeddings, E.KEYED_VECTOR_TYPES):
            raise ValueError('The parameter `input_embeddings` is wrong! Expected `{0}`, got `{1}`.'.format(
                E.expected_type(), type(self.input_embeddings)))
        if self.output_embeddings is not None:
            if not isinstance(self.output_embeddings, E.KEYED_VECTOR_TYPES):
                raise ValueError('The parameter `output_embeddings` is wrong! Expected `{0}`, got `{1}`.'.format(
                    E.expected_type(), type(self.output_embeddings)))

    # ---- static helpers ------------------------------------------------------------------------------------------
    @staticmethod
    def postprocess_reconstructed_word_vectors(word_vectors: np.ndarray):
        """EOS truncation + renormalisation (:542-565) through `c1v_eos_truncate_normalize` (the dtype is kept)."""
        if len(word_vectors.shape) != 3:
            raise ValueError('The `word_vectors` parameter is wrong! Expected 3-D array, got {0}-D array!'.format(
                len(word_vectors.shape)))
        from .engine import postprocess_word_vectors
        return postprocess_word_vectors(word_vectors, Conv1dTextVAE.device)[0]

    @staticmethod
    def find_best_words(word_vector: np.ndarray, embeddings_model, n: int, special_symbols: tuple = None):
        """Nearest words of one reconstructed vector (:580-642).  Returns None when the end-of-sentence flag wins, [] when
        the unknown-word flag wins, [(special symbol, distance)] when a special-symbol flag wins, otherwise the `n`
        nearest vocabulary words; `similar_by_vector` of the shipped embedding container is the HIP GEMM + top-n."""
        emb = embeddings_model.vector_size
        vector_size = Conv1dTextVAE.calc_vector_size(embeddings_model, special_symbols)
        norm_value = np.linalg.norm(word_vector[:emb])
        if norm_value < EPSILON:
            norm_value = 1.0
        similar = embeddings_model.similar_by_vector(word_vector[:emb] / norm_value, topn=n)

        def padded_unit(word):
            v = np.zeros((vector_size,), dtype=np.float32)
            v[0:emb] = embeddings_model[word]
            nv = np.linalg.norm(v)
            return v / nv if nv > 0.0 else v

        def flag(slot):
            v = np.zeros((vector_size,), dtype=np.float32)
            v[slot] = 1.0
            return v

        to_end = distance.cosine(word_vector, flag(vector_size - 1))
        to_unknown = distance.cosine(word_vector, flag(vector_size - 2))
        to_best = distance.cosine(word_vector, padded_unit(similar[0][0]))
        special = None
        if (special_symbols is not None) and (len(special_symbols) > 0):
            d = np.array([distance.cosine(word_vector, flag(emb + i)) for i in range(len(special_symbols))], dtype=np.float32)
            i = int(d.argmin())
            special = [(special_symbols[i], float(d[i]))]
        if to_end < to_unknown:
            if to_end < to_best:
                return None if (special is None or to_end < special[0][1]) else special
            return similar
        if to_unknown < to_best:
            return [] if (special is None or to_unknown < special[0][1]) else special
        if special is not None:
            return special if special[0][1] < to_best else similar
        rescored = [(word, float(distance.cosine(word_vector, padded_unit(word)))) for word, _ in similar]
        rescored.sort(key=lambda it: (it[1], it[0]))
        return rescored

    @staticmethod
    def find_best_texts(variants_of_text: List[tuple], ntop: int) -> List[str]:
        """The best text and `ntop - 1` single-word substitutions in order of increasing distance (:644-662)."""
        current = [variants[0][0] for variants in variants_of_text]
        texts = [' '.join(current)]
        alternatives = [((word_idx, variant_idx), variants[variant_idx][1])
                        for word_idx, variants in enumerate(variants_of_text) for variant_idx in range(1, len(variants))]
        alternatives.sort(key=lambda it: (it[1], it[0][0], it[0][1]))
        for (word_idx, variant_idx), _ in alternatives[:max(ntop - 1, 0)]:
            current[word_idx] = variants_of_text[word_idx][variant_idx][0]
            texts.append(' '.join(current))
        return texts

    @staticmethod
    def copy_embeddings(src):
        tmp_fasttext_name = Conv1dTextVAE.get_temp_name()
        try:
            src.save(tmp_fasttext_name)
            res = type(src).load(tmp_fasttext_name)
        finally:
            Conv1dTextVAE.remove_fasttext_files(tmp_fasttext_name)
        return res

    @staticmethod
    def check_texts_param(param_value, param_name: str):
        """A list, a tuple or a 1-D array of string-like items; the reference's messages (:675-688)."""
        kinds = (list, tuple, np.ndarray)
        if not isinstance(param_value, kinds):
            raise ValueError('The parameter `{0}` is wrong! Expected `{1}`, `{2}` or 1-D `{3}`, got `{4}`.'.format(
                param_name, *kinds, type(param_value)))
        n_dims = param_value.ndim if isinstance(param_value, np.ndarray) else 1
        if n_dims != 1:
            raise ValueError('The parameter `{0}` is wrong! Expected 1-D array, got {1}-D array.'.format(param_name, n_dims))
        for idx, item in enumerate(param_value):
            if not (hasattr(item, 'split') and hasattr(item, 'strip')):
                raise ValueError('Item {0} of the parameter `{1}` is wrong! This item is not string!'.format(idx, param_name))

    @staticmethod
    def check_params(**params):
        """Hyper-parameter validation with the reference's messages and order (:691-811)."""
        def need(name):
            if name not in params:
                raise ValueError('The parameter `{0}` is not defined!'.format(name))
            return params[name]

        def wrong_type(name, expected, value):
            return ValueError('The parameter `{0}` is wrong! Expected `{1}`, got `{2}`.'.format(name, expected, type(value)))

        def not_positive(name, value):
            return ValueError('The parameter `{0}` is wrong! Expected a positive value, but {1} is not positive.'.format(
                name, value))

        for name in ('warm_start', 'use_batch_norm', 'use_attention', 'verbose'):
            value = need(name)
            if (not isinstance(value, bool)) and (not isinstance(value, int)):
                raise wrong_type(name, type(True), value)
        for name in ('batch_size', 'max_epochs', 'latent_dim'):
            value = need(name)
            if not isinstance(value, int):
                raise wrong_type(name, type(10), value)
            if value <= 0:
                raise not_positive(name, value)
        for name in ('input_text_size', 'output_text_size', 'output_onehot_size'):
            value = need(name)
            if value is not None:
                if not isinstance(value, int):
                    raise wrong_type(name, type(10), value)
                if value <= 0:
                    raise not_positive(name, value)
        value = need('n_filters')
        if (not isinstance(value, int)) and (not isinstance(value, tuple)):
            raise ValueError('The parameter `n_filters` is wrong! Expected `{0}` or `{1}`, got `{2}`.'.format(
                type(10), type((1, 2)), type(value)))
        if isinstance(value, int):
            if value <= 0:
                raise not_positive('n_filters', value)
        else:
            if len(value) < 1:
                raise ValueError('The parameter `n_filters` is wrong! Expected a nonempty sequence of integers.')
            for idx in range(len(value)):
                if value[idx] <= 0:
                    raise ValueError('Item {0} of the parameter `n_filters` is wrong! Expected a positive value, '
                                     'but {1} is not positive.'.format(idx, value[idx]))
        for name in ('kernel_size', 'n_recurrent_units'):
            value = need(name)
            if not isinstance(value, int):
                raise wrong_type(name, type(10), value)
            if value <= 0:
                raise not_positive(name, value)
        value = need('validation_fraction')
        if not isinstance(value, float):
            raise wrong_type('validation_fraction', type(10.5), value)
        if (value <= 0) or (value >= 1.0):
            raise ValueError('The parameter `validation_fraction` is wrong! Expected a positive value between 0.0 and '
                             '1.0, but {0} does not correspond to this condition.'.format(
                                 Conv1dTextVAE.float_to_string(value)))
        value = need('lr')
        if not isinstance(value, float):
            raise wrong_type('lr', type(10.5), value)
        if value <= 0:
            raise not_positive('lr', value)

    @staticmethod
    def calc_vector_size(embeddings, special_symbols):
        vector_size = embeddings.vector_size + 2
        if special_symbols is not None:
            vector_size += len(special_symbols)
        return vector_size

    @staticmethod
    def tokenize(src: str, bounds_of_words: List[Tuple[int, int]]) -> tuple:
        return tuple(w for w in (src[a:b].lower() for a, b in bounds_of_words) if len(w) > 0)

    @staticmethod
    def texts_to_data(input_texts, batch_size: int, max_text_size: int, tokenizer: BaseTokenizer, fasttext_model,
                      special_symbols: tuple = None):
        """Inference featuriser (:829-866): yields float32 [batch_size, max_text_size, vector_size]; the last batch is
        padded by repeating the final text.  Slots: special symbol i -> emb + i, unknown word -> vector_size - 2,
        end of sentence -> vector_size - 1 (the reference's inference layout)."""
        n_batches = int(math.ceil(len(input_texts) / batch_size))
        emb = fasttext_model.vector_size
        vector_size = Conv1dTextVAE.calc_vector_size(fasttext_model, special_symbols)
        for batch_ind in range(n_batches):
            input_data = np.zeros((batch_size, max_text_size, vector_size), dtype=np.float32)
            for row in range(batch_size):
                text = input_texts[min(batch_ind * batch_size + row, len(input_texts) - 1)]
                words = Conv1dTextVAE.tokenize(text, tokenizer.tokenize_into_words(text))
                for time_idx, token in enumerate(words[:max_text_size]):
                    if (special_symbols is not None) and (token in special_symbols):
                        input_data[row, time_idx, emb + special_symbols.index(token)] = 1.0
                        continue
                    try:
                        word_vector = fasttext_model[token]
                    except Exception:                                   # noqa: BLE001 - as the reference (:853)
                        word_vector = None
                    if word_vector is None:
                        input_data[row, time_idx, vector_size - 2] = 1.0
                    else:
                        vector_norm = np.linalg.norm(word_vector)
                        if vector_norm < EPSILON:
                            vector_norm = 1.0
                        input_data[row, time_idx, 0:emb] = word_vector / vector_norm
                if len(words) < max_text_size:
                    input_data[row, len(words), vector_size - 1] = 1.0
            yield input_data

    @staticmethod
    def get_temp_name():
        fp = tempfile.NamedTemporaryFile(delete=True)
        file_name = fp.name
        fp.close()
        return file_name

    @staticmethod
    def remove_fasttext_files(file_name: str):
        if os.path.isfile(file_name):
            os.remove(file_name)
        dir_name, base_name = os.path.dirname(file_name), os.path.basename(file_name)
        for cur in os.listdir(dir_name):
            if cur.startswith(base_name) and cur.endswith('.npy') and os.path.isfile(os.path.join(dir_name, cur)):
                os.remove(os.path.join(dir_name, cur))

    @staticmethod
    def float_to_string(value: float, precision: int = 6) -> str:
        """Fixed-point text without leading / trailing zeros (:888-913): 0.50670 -> '0.5067', 3.0 -> '3'."""
        if not isinstance(value, float):
            return str(value)
        res = '{0:.{1}f}'.format(value, precision)
        stripped = res.lstrip('0')
        if len(stripped) == 0:
            return '0'
        if stripped[0] == '.':
            stripped = '0' + stripped
        if '.' in stripped:
            stripped = stripped.rstrip('0')
            if stripped.endswith('.'):
                stripped = stripped[:-1]
        return stripped

    @staticmethod
    def quantize_word_vectors(word_vectors: np.ndarray, max_vocabulary_size: int, verbose: bool):
        """`output_onehot_size`: merge near-duplicate words of the target vocabulary into clusters (reference :916-1007).

        The reference builds an Annoy index (angular metric) over the unit word vectors, takes every word's
        n = min(3 * max(1, N // max_vocabulary_size), N - 1) nearest neighbours (itself included, distance 0) as a sparse
        distance matrix, runs scikit-learn's DBSCAN on it (metric 'precomputed', eps = the mean of all stored distances,
        min_samples = max(1, N // (4 * max_vocabulary_size))), turns every noise point into a cluster of its own and
        returns (cluster id per word, L2-normalised cluster centroids).

        Here the neighbours are EXACT: one fp32 MFMA GEMM of the vocabulary against itself with a fused per-row top-n
        (`c1v_vocab_topn`, the nearest-word kernel of SURVEY.md row a10-ii), angular distance sqrt(2 - 2 cos) as Annoy
        defines it.  Annoy is an approximate, randomly seeded index (and is not installable here), so its neighbour lists
        -- and with them the clusters -- are not reproducible even between two runs of the reference: parity unpinned by
        construction; exact neighbours are the limit its `search_k=-1` query approximates.  DBSCAN itself is scikit-learn's,
        as in the reference."""
        from scipy.sparse import csr_matrix
        from sklearn.cluster import DBSCAN
        from .engine import VocabTopN
        vectors = np.ascontiguousarray(word_vectors, dtype=np.float32)
        n_words = vectors.shape[0]
        n_neighbours = min(3 * max(1, int(n_words // max_vocabulary_size)), n_words - 1)
        if verbose:
            print('')
            print('----------------------------------------')
            print('Calculation of neighbourhood matrix is started...')
            print('----------------------------------------')
        start_time = time.time()
        score, index = VocabTopN(vectors, device=Conv1dTextVAE.device).run(vectors, n_neighbours)
        # every word is its own nearest neighbour at distance 0 (Annoy's get_nns_by_item returns the item first); a
        # duplicate row could tie with it, so the diagonal is written explicitly
        rows = np.repeat(np.arange(n_words, dtype=np.int32), n_neighbours)
        dist = np.sqrt(np.maximum(2.0 - 2.0 * score.astype(np.float64), 0.0)).astype(np.float32)
        has_self = (index == np.arange(n_words)[:, None]).any(axis=1)
        index[~has_self, -1] = np.nonzero(~has_self)[0]
        dist[index == np.arange(n_words)[:, None]] = 0.0
        if verbose:
            print('{0:.3f} seconds:\t100% of vectors has been processed...'.format(time.time() - start_time))
            print('Number of all elements is {0}.'.format(n_words * n_words))
            print('Part of nonzero elements is {0:.3%}.'.format(dist.size / float(n_words * n_words)))
        neighbourhood = csr_matrix((dist.reshape(-1), (rows, index.reshape(-1).astype(np.int32))), shape=(n_words, n_words))
        clustering = DBSCAN(n_jobs=1, min_samples=max(1, int(n_words // (max_vocabulary_size * 4))), metric='precomputed',
                            eps=float(np.mean(dist)))
        if verbose:
            print('')
            print('----------------------------------------')
            print('DBSCAN clustering with scikit-learn is started...')
            print('----------------------------------------')
            print('n_samples = {0}'.format(n_words))
            print('DBSCAN.min_samples = {0}'.format(clustering.min_samples))
            print('DBSCAN.eps = {0}'.format(clustering.eps))
        labels = clustering.fit_predict(neighbourhood)
        n_clusters = len(set(labels.tolist()) - {-1})
        noise = labels < 0
        # noise points become clusters of their own, numbered after the "good" ones in word order
        word_clusters = np.where(noise, n_clusters + np.cumsum(noise) - 1, labels).astype(np.int64)
        n_all = n_clusters + int(noise.sum())
        if verbose:
            print('DBSCAN training is finished...')
            print('Number of "good" clusters is {0}.'.format(n_clusters))
            print('Number of all clusters (including noisy data) is {0}.'.format(n_all))
            print('')
            print('Quantization with DBSCAN is started...')
        centers = np.zeros((n_all, vectors.shape[1]), dtype=np.float32)
        np.add.at(centers, word_clusters, vectors)
        centers /= np.bincount(word_clusters, minlength=n_all).astype(np.float32)[:, None]
        centers /= np.linalg.norm(centers, axis=1, keepdims=True)
        if verbose:
            print('Quantization with DBSCAN is finished...')
            print('')
        return [int(c) for c in word_clusters], centers

    @staticmethod
    def get_vocabulary_and_word_vectors_from_fasttext(all_texts, fasttext_vectors, special_symbols,
                                                      max_vocabulary_size: int = None, verbose: bool = False):
        """Word -> row over the corpus in first-occurrence order, rows L2-normalised; words without a vector or with a
        zero vector are left out (:1010-1040)."""
        vocabulary, rows = dict(), []
        for cur_text in all_texts:
            for cur_word in cur_text:
                if len(cur_word) == 0 or cur_word in vocabulary:
                    continue
                if (special_symbols is not None) and (cur_word in special_symbols):
                    continue
                try:
                    word_vector = fasttext_vectors[cur_word]
                except Exception:                                       # noqa: BLE001 - as the reference (:1022)
                    continue
                if word_vector is None:
                    continue
                vector_norm = np.linalg.norm(word_vector)
                if vector_norm > EPSILON:
                    vocabulary[cur_word] = len(rows)
                    rows.append(np.asarray(word_vector, dtype=np.float32) / vector_norm)
        word_vectors = np.zeros((len(rows), fasttext_vectors.vector_size), dtype=np.float32)
        if rows:
            word_vectors[:] = np.vstack(rows)
        if (max_vocabulary_size is not None) and (max_vocabulary_size < word_vectors.shape[0]):
            word_clusters, word_vectors = Conv1dTextVAE.quantize_word_vectors(word_vectors, max_vocabulary_size, verbose)
            for cur_word in vocabulary:
                vocabulary[cur_word] = word_clusters[vocabulary[cur_word]]
        return vocabulary, word_vectors

    @staticmethod
    def prepare_vocabulary_and_word_vectors(all_texts, fasttext_vectors, special_symbols, max_vocabulary_size: int = None,
                                            verbose: bool = False):
        """The matrix W of the loss (:1043-1081): [n_words + 1 + n_special + 1, emb + 2 + n_special] with unit rows: word
        rows, the unknown-word row (slot emb), special-symbol rows (slot emb + 1 + i), the end-of-sentence row (last
        slot); `vocabulary['']` is the end-of-sentence row."""
        src_vocabulary, src_vectors = Conv1dTextVAE.get_vocabulary_and_word_vectors_from_fasttext(
            all_texts, fasttext_vectors, special_symbols, max_vocabulary_size, verbose)
        emb = fasttext_vectors.vector_size
        n_words = src_vectors.shape[0]
        n_special = 0 if special_symbols is None else len(special_symbols)
        vector_size = Conv1dTextVAE.calc_vector_size(fasttext_vectors, special_symbols)
        word_vectors = np.zeros((n_words + n_special + 2, vector_size), dtype=np.float32)
        word_vectors[0:n_words, 0:emb] = src_vectors
        word_vectors[n_words, emb] = 1.0
        for i in range(n_special):
            word_vectors[n_words + 1 + i, emb + 1 + i] = 1.0
        word_vectors[word_vectors.shape[0] - 1, vector_size - 1] = 1.0
        vocabulary = dict()
        for cur_text in all_texts:
            for cur_word in cur_text:
                if len(cur_word) == 0 or cur_word in vocabulary:
                    continue
                if n_special > 0 and cur_word in special_symbols:
                    vocabulary[cur_word] = n_words + 1 + special_symbols.index(cur_word)
                else:
                    vocabulary[cur_word] = src_vocabulary.get(cur_word, n_words)
        vocabulary[''] = word_vectors.shape[0] - 1
        return vocabulary, word_vectors

    # ---- pickling (:1112-1183) -----------------------------------------------------------------------------
    def __getstate__(self):
        state = self.get_params(deep=False)
        state.pop('input_embeddings', None)
        state.pop('output_embeddings', None)
        if all(hasattr(self, it) for it in FITTED_ATTRIBUTES):
            for name in ('input_text_size_', 'output_text_size_', 'input_vector_size_', 'output_vector_size_',
                         'output_text_size_in_characters_'):
                state[name] = getattr(self, name)
            state['target_char_index_'] = copy.deepcopy(self.target_char_index_)
            state['reverse_target_char_index_'] = copy.deepcopy(self.reverse_target_char_index_)
            state['weights_'] = {
                'vae_encoder': _npz_bytes(self.vae_encoder_.get_weights()),
                'vae_decoder': _npz_bytes(self.vae_decoder_.get_weights()),
                'seq2seq_encoder': _npz_bytes(self.generator_encoder_.get_weights()),
                'seq2seq_decoder': _npz_bytes(self.generator_decoder_.get_weights()),
            }
        return state

    def __setstate__(self, state):
        if not isinstance(state, dict):
            raise ValueError('`state` is wrong! Expected {0}.'.format(type({0: 1})))
        state['input_embeddings'] = None
        state['output_embeddings'] = None
        self.check_params(**state)
        self._release_models()
        is_fitted = all(it in state for it in ['input_text_size_', 'output_text_size_', 'input_vector_size_',
                                               'output_vector_size_', 'weights_', 'output_text_size_in_characters_',
                                               'target_char_index_', 'reverse_target_char_index_'])
        self.set_params(**state)
        if is_fitted:
            if not isinstance(state['weights_'], dict):
                raise ValueError('Weights are wrong! Expected `{0}`, got `{1}`.'.format(type({'a': 1, 'b': 2}),
                                                                                        type(state['weights_'])))
            for key in ('vae_encoder', 'vae_decoder', 'seq2seq_encoder', 'seq2seq_decoder'):
                if key not in state['weights_']:
                    raise ValueError('Weights are wrong! The key `{0}` is not found.'.format(key))
            self.input_text_size_ = state['input_text_size_']
            self.output_text_size_ = state['output_text_size_']
            self.output_text_size_in_characters_ = state['output_text_size_in_characters_']
            self.target_char_index_ = copy.deepcopy(state['target_char_index_'])
            self.reverse_target_char_index_ = copy.deepcopy(state['reverse_target_char_index_'])
            self.input_vector_size_ = state['input_vector_size_']
            self.output_vector_size_ = state['output_vector_size_']
            vae_weights = dict(_npz_load(state['weights_']['vae_encoder']))
            vae_weights.update(_npz_load(state['weights_']['vae_decoder']))
            runtime, generator = self._create_model(self.input_vector_size_, self.output_vector_size_, seed=0,
                                                    vae_weights=vae_weights)
            generator.set_weights(_npz_load(state['weights_']['seq2seq_encoder']))
            generator.set_weights(_npz_load(state['weights_']['seq2seq_decoder']))
            self._vae_runtime = runtime
            self._generator = generator
            self.vae_encoder_ = VAEEncoderModel(runtime)
            self.vae_decoder_ = VAEDecoderModel(runtime)
            self.generator_encoder_ = generator.encoder_model()
            self.generator_decoder_ = generator.decoder_model()

    # ---- model construction (:1185-1343) -----------------------------------------------------------------------
    def _create_model(self, input_vector_size: int, output_vector_size: int, seed: int,
                      encoder_weights: Optional[Dict[str, np.ndarray]] = None,
                      vae_weights: Optional[Dict[str, np.ndarray]] = None):
        from .seq2seq import Seq2SeqGenerator
        geom = Geometry(self.input_text_size_, self.output_text_size_, input_vector_size, output_vector_size,
                        self.n_filters, self.kernel_size, self.latent_dim, use_batch_norm=bool(self.use_batch_norm))
        weights = init_keras_weights(geom, seed)
        if vae_weights is not None:
            missing = set(k for k in weights if not k.startswith(UNSAVED_PREFIXES)) - set(vae_weights)
            if missing:
                raise ValueError('Weights are wrong! Missing arrays: {0}.'.format(sorted(missing)))
            weights = {k: np.asarray(vae_weights[k], dtype=np.float32) if k in vae_weights else weights[k] for k in weights}
        if encoder_weights is not None:
            for k, v in encoder_weights.items():
                if tuple(v.shape) != tuple(weights[k].shape):
                    raise ValueError('Input embeddings do not correspond to the neural network!')
                weights[k] = np.asarray(v, dtype=np.float32)
        runtime = _VAERuntime(geom, self.batch_size, self.compute_dtype, self.device, weights, seed=seed)
        try:
            import torch
            gen_device = self.device if torch.cuda.is_available() else 'cpu'
        except Exception:                                               # noqa: BLE001
            gen_device = 'cpu'
        generator = Seq2SeqGenerator(self.output_text_size_, output_vector_size - 1, len(self.target_char_index_),
                                     self.n_recurrent_units, use_attention=bool(self.use_attention), lr=self.lr,
                                     device=gen_device, seed=seed + 1)
        return runtime, generator

    @staticmethod
    def _print_summary(geom: Geometry, generator):
        def table(title, rows):
            print('')
            print(title)
            total = 0
            for name, shape in rows:
                n = int(np.prod(shape))
                total += n
                print('  {0:<44s} {1:<24s} {2:>12d}'.format(name, str(tuple(shape)), n))
            print('  total parameters: {0}'.format(total))
        shapes = geom.keras_shapes()
        table('ENCODER:', [(k, v) for k, v in shapes.items() if k.startswith(ENCODER_PREFIXES)])
        table('DECODER:', [(k, v) for k, v in shapes.items() if not k.startswith(ENCODER_PREFIXES + UNSAVED_PREFIXES)])
        table('SEQUENCE-TO-SEQUENCE:', [(k, tuple(v.shape)) for k, v in generator.p.items()])
