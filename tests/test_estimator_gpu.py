"""End-to-end checks of the drop-in estimator on the GPU: the reference's integration tests
(tests/test_conv1d_text_vae.py:1114-1178, :1180-1325, :1373-1429, :1483-1619) replayed with the synthetic FastText
provider on the reference's own test corpus (tests/golden/testdata.csv = the reference's tests/testdata/testdata.csv).
As in the reference these pin attribute presence, shapes, types and non-empty strings; numerical parity of the kernels
underneath is pinned in test_engine_gpu.py / test_kernels_gpu.py against the oracle.
"""
import pickle

import numpy as np
import pytest

from oracle import vae_oracle as O
from test_estimator_cpu import FITTED, HYPER, load_testdata

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _seed_global_rng():
    """fit() draws its model seed, validation split and shuffles from NumPy's global generator (as the reference relies
    on Keras / NumPy global state); seeding it per test makes the outcome independent of the test order."""
    np.random.seed(20240607)
    yield


@pytest.fixture(scope='module')
def corpus():
    return load_testdata(max_target_words=14)


def test_fit_full_length_texts(fasttext_model):
    """the uncut corpus (full texts of several hundred tokens / ~1500 characters), two epochs"""
    from conv1d_text_vae_amd import Conv1dTextVAE
    X, y = load_testdata()
    vae = Conv1dTextVAE(fasttext_model, fasttext_model, max_epochs=2, batch_size=2, n_filters=(32, 16))
    res = vae.fit(X, y)
    _assert_fitted(res)
    assert res.output_text_size_ > 100 and res.output_text_size_in_characters_ > 1000
    out = res.predict(X[:2])
    assert len(out) == 2 and all(isinstance(s, str) and len(s) > 0 for s in out)


@pytest.fixture(scope='module')
def fasttext_model(corpus):
    from conv1d_text_vae_amd import Conv1dTextVAE, DefaultTokenizer
    from conv1d_text_vae_amd import embeddings as E
    tok = DefaultTokenizer()
    words = []
    for t in corpus[0] + corpus[1]:
        words += list(Conv1dTextVAE.tokenize(t, tok.tokenize_into_words(t)))
    words += ['мама', 'мыть', 'рама', 'нейросеть']
    return E.synthetic_fasttext(words, vector_size=40, bucket=2000, seed=7)


@pytest.fixture()
def text_vae(fasttext_model):
    from conv1d_text_vae_amd import Conv1dTextVAE
    return Conv1dTextVAE(input_embeddings=fasttext_model, output_embeddings=fasttext_model, max_epochs=10, batch_size=2)


def _assert_fitted(res):
    from conv1d_text_vae_amd import Conv1dTextVAE
    assert isinstance(res, Conv1dTextVAE)
    for name in HYPER + FITTED:
        assert hasattr(res, name), name


def test_fit_positive01(text_vae, corpus):
    text_vae.verbose = 2
    text_vae.use_attention = False
    res = text_vae.fit(corpus[0], corpus[1])
    _assert_fitted(res)
    assert res.input_vector_size_ == 42 and res.output_vector_size_ == 42
    assert len(res.vae_history_) >= 1 and np.isfinite(res.vae_history_[-1]['val_loss'])
    assert len(res.seq2seq_history_) >= 1 and np.isfinite(res.seq2seq_history_[-1]['val_loss'])


def test_fit_positive03_warm_start_keeps_encoder(text_vae, corpus):
    res = text_vae.fit(corpus[0])
    enc_before = res.vae_encoder_.get_weights()
    # `vae_encoder_` is Model(encoder_input, z_mean) (:1308): the z_log_var head is not part of it ...
    assert not any(k.startswith('z_log_var') for k in enc_before) and any(k.startswith('z_mean') for k in enc_before)
    lv_before = res._vae_runtime.weights()['z_log_var/kernel'].copy()
    res.warm_start = True
    res.fit(corpus[0], corpus[1])
    _assert_fitted(res)
    enc_after = res.vae_encoder_.get_weights()
    assert set(enc_before) == set(enc_after)
    # ... so a warm start gives that head a fresh initialisation (and freezes it with the rest of the encoder, :1245)
    assert not np.array_equal(lv_before, res._vae_runtime.weights()['z_log_var/kernel'])
    for k in enc_before:                                   # encoder layers are frozen in a warm start (:1228, :1244-1245)
        np.testing.assert_array_equal(enc_before[k], enc_after[k])


def test_loss_decreases_over_fit(fasttext_model, corpus):
    from conv1d_text_vae_amd import Conv1dTextVAE
    vae = Conv1dTextVAE(fasttext_model, fasttext_model, max_epochs=12, batch_size=2, latent_dim=16)
    vae.fit(corpus[0])
    hist = vae.vae_history_
    assert hist[-1]['loss'] < hist[0]['loss']


def test_transform_positive01(text_vae, corpus):
    res = text_vae.fit_transform(corpus[0], corpus[1])
    assert isinstance(res, np.ndarray) and res.dtype == np.float32
    assert res.shape == (len(corpus[0]), text_vae.latent_dim)
    assert np.isfinite(res).all()
    # transform == vae_encoder_.predict_on_batch on the featurised texts, row by row, whatever the batch split is
    again = text_vae.transform(corpus[0][:3])
    np.testing.assert_allclose(again, res[:3], rtol=1e-5, atol=1e-6)


def test_transform_negative02(text_vae, corpus):
    texts = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9], [10, 11, 12]])
    text_vae.fit(corpus[0])
    with pytest.raises(ValueError, match='Expected 1-D array, got 2-D array'):
        text_vae.transform(texts)
    with pytest.raises(ValueError, match='Expected 1-D array, got 2-D array'):
        text_vae.predict(texts)
    text_vae.input_embeddings = None
    with pytest.raises(ValueError, match='The parameter `input_embeddings` is wrong!'):
        text_vae.predict(corpus[0])


def test_predict_positive01(text_vae, corpus):
    batch_size = 3
    while (len(corpus[0]) % batch_size) == 0:
        batch_size += 1
    text_vae.batch_size = batch_size
    text_vae.max_epochs *= 2
    res = text_vae.fit_predict(corpus[0] + corpus[0], corpus[1] + corpus[1])
    assert isinstance(res, list) and len(res) == 2 * len(corpus[0])
    for s in res:
        assert isinstance(s, str) and len(s.strip()) > 0
    assert isinstance(text_vae.predict(tuple(corpus[0][:2])), tuple)
    arr = text_vae.predict(np.array(corpus[0][:2], dtype=object))
    assert isinstance(arr, np.ndarray) and arr.shape == (2,)


def test_predict_matches_model_facades(text_vae, corpus):
    """predict()'s device-side pipeline equals the reference's formulation through the four model objects (:437-470)."""
    from conv1d_text_vae_amd import Conv1dTextVAE
    text_vae.fit(corpus[0], corpus[1])
    X = corpus[0][:4]
    got = text_vae.predict(X)
    tok = text_vae.tokenizer
    want = []
    ci, rci = text_vae.target_char_index_, text_vae.reverse_target_char_index_
    for data in Conv1dTextVAE.texts_to_data(X, text_vae.batch_size, text_vae.input_text_size_, tok, text_vae.input_embeddings, None):
        words = Conv1dTextVAE.postprocess_reconstructed_word_vectors(
            text_vae.vae_decoder_.predict_on_batch(text_vae.vae_encoder_.predict_on_batch(data)))
        words = np.nan_to_num(words, nan=0.0, posinf=0.0, neginf=0.0)
        state = text_vae.generator_encoder_.predict_on_batch(words)
        B = data.shape[0]
        target_seq = np.zeros((B, 1, len(ci)), dtype=np.float32)
        target_seq[:, 0, ci['<BOS>']] = 1.0
        stop, decoded = [False] * B, [[] for _ in range(B)]
        while not all(stop):
            out, state = text_vae.generator_decoder_.predict_on_batch([target_seq, state])
            idx = np.argmax(out[:, -1, :], axis=1)
            for b in range(B):
                if stop[b]:
                    continue
                ch = rci[int(idx[b])]
                decoded[b].append(ch)
                if ch == '<EOS>' or len(decoded[b]) > text_vae.output_text_size_in_characters_:
                    stop[b] = True
                target_seq[b, 0, :] = 0.0
                target_seq[b, 0, idx[b]] = 1.0
        want += [''.join(d) for d in decoded]
    assert got == want[:len(X)]


def test_serialize_fitted(text_vae, corpus, tmp_path):
    from conv1d_text_vae_amd import Conv1dTextVAE
    text_vae.fit(corpus[0], corpus[1])
    name = str(tmp_path / 'model.pkl')
    with open(name, 'wb') as fp:
        pickle.dump(text_vae, fp)
    with open(name, 'rb') as fp:
        res = pickle.load(fp)
    assert isinstance(res, Conv1dTextVAE) and res is not text_vae
    _assert_fitted(res)
    assert res.input_embeddings is None and res.output_embeddings is None
    for name in ('n_filters', 'kernel_size', 'batch_size', 'max_epochs', 'latent_dim', 'n_recurrent_units', 'warm_start',
                 'verbose', 'use_attention', 'use_batch_norm', 'input_text_size', 'output_text_size',
                 'validation_fraction', 'lr'):
        assert getattr(res, name) == getattr(text_vae, name)
    res.input_embeddings = text_vae.input_embeddings
    res.output_embeddings = text_vae.output_embeddings
    X1 = text_vae.transform(corpus[0])
    X2 = res.transform(corpus[0])
    assert X1.shape == X2.shape and X1.dtype == X2.dtype
    np.testing.assert_array_equal(X1, X2)                 # same weights, same kernels: bit-identical
    assert res.predict(corpus[0][:3]) == text_vae.predict(corpus[0][:3])


# ---- static helpers that run on the GPU -----------------------------------------------------------------------
@pytest.mark.parametrize('true_end_idx', [11, 1])
def test_postprocess_reconstructed_word_vectors_positive(true_end_idx):
    """reference tests :1544-1605 (float64 in, float64 out, 7 decimal places)"""
    from conv1d_text_vae_amd import Conv1dTextVAE
    rng = np.random.RandomState(true_end_idx)
    src = np.hstack((rng.uniform(-1.0, 1.0, (20, 6)), rng.uniform(-3.0, -2.0, (20, 1))))
    src[true_end_idx] = 0.0
    src[true_end_idx][-1] = 1.0
    if true_end_idx == 1:
        src[0][-1] = 1.5                                   # EOS at step 0 still keeps one word: end = max(t, 1)
    true = np.zeros((20, 6))
    for i in range(true_end_idx):
        true[i] = src[i][:6] / np.linalg.norm(src[i][:6])
    got = Conv1dTextVAE.postprocess_reconstructed_word_vectors(src.reshape(1, 20, 7))
    assert isinstance(got, np.ndarray) and got.shape == (1, 20, 6) and got.dtype == src.dtype
    assert np.abs(got[0] - true).max() < 5e-8
    got32 = Conv1dTextVAE.postprocess_reconstructed_word_vectors(src.reshape(1, 20, 7).astype(np.float32))
    assert got32.dtype == np.float32 and np.abs(got32[0] - true).max() < 1e-6


def test_postprocess_matches_oracle_on_batches():
    from conv1d_text_vae_amd import Conv1dTextVAE
    rng = np.random.RandomState(3)
    y = rng.standard_normal((37, 19, 45)).astype(np.float32)
    y[::3, 5, -1] = 10.0
    y[1, 0, -1] = 10.0
    np.testing.assert_allclose(Conv1dTextVAE.postprocess_reconstructed_word_vectors(y),
                               O.postprocess_reconstructed_word_vectors(y), rtol=1e-6, atol=1e-7)


def test_find_best_words_and_texts(fasttext_model):
    """reference tests :1114-1178"""
    from conv1d_text_vae_amd import Conv1dTextVAE
    emb = fasttext_model.vector_size
    v = np.zeros((emb + 2,), dtype=np.float32)
    v[:emb] = fasttext_model['нейросеть']
    best = Conv1dTextVAE.find_best_words(v, fasttext_model, 3)
    assert isinstance(best, list) and len(best) == 3
    for w, d in best:
        assert isinstance(w, str) and isinstance(d, float) and d >= -1e-6
    assert best[0][0] == 'нейросеть' and abs(best[0][1]) < 1e-5
    assert all(best[i - 1][1] <= best[i][1] for i in range(1, 3))
    # the oracle's restatement over the explicit table gives the same answer
    ref = O.find_best_words(v, fasttext_model.vectors_norm, fasttext_model.index2word, fasttext_model.vectors, 3)
    assert [w for w, _ in best] == [w for w, _ in ref]
    np.testing.assert_allclose([d for _, d in best], [d for _, d in ref], atol=1e-5)
    v = np.zeros((emb + 2,), dtype=np.float32)
    v[emb + 1] = 1.0
    assert Conv1dTextVAE.find_best_words(v, fasttext_model, 3) is None               # end of sentence wins
    v = np.zeros((emb + 2,), dtype=np.float32)
    v[emb] = 1.0
    assert Conv1dTextVAE.find_best_words(v, fasttext_model, 3) == []                 # unknown word wins
    v = np.zeros((emb + 4,), dtype=np.float32)
    v[emb + 1] = 1.0
    best = Conv1dTextVAE.find_best_words(v, fasttext_model, 3, ('\t', '\n'))
    assert len(best) == 1 and best[0][0] == '\n' and abs(best[0][1]) < 1e-6
    variants = []
    for w in ('мама', 'мыть', 'рама'):
        v = np.zeros((emb + 2,), dtype=np.float32)
        v[:emb] = fasttext_model[w]
        variants.append(tuple(Conv1dTextVAE.find_best_words(v, fasttext_model, 5)))
    texts = Conv1dTextVAE.find_best_texts(variants, 5)
    assert isinstance(texts, list) and len(texts) == 5 and texts[0] == 'мама мыть рама'


def test_similar_by_vector_topn_matches_numpy(fasttext_model):
    rng = np.random.RandomState(0)
    q = rng.standard_normal((9, fasttext_model.vector_size)).astype(np.float32)
    res = fasttext_model.similar_by_vectors(q, topn=7)
    sims = (q / np.linalg.norm(q, axis=1, keepdims=True)) @ fasttext_model.vectors_norm.T
    for m in range(9):
        order = np.argsort(-sims[m], kind='stable')[:7]
        assert [w for w, _ in res[m]] == [fasttext_model.index2word[i] for i in order]
        np.testing.assert_allclose([s for _, s in res[m]], sims[m][order], atol=2e-6)


def test_sequence_for_seq2seq_with_fake_vae(fasttext_model):
    """reference tests :278-512 with the identity FakeVAE (:270-275): teacher-forcing shift and BOS / EOS placement"""
    from conv1d_text_vae_amd import Conv1dTextVAE, DefaultTokenizer, SequenceForSeq2Seq

    class FakeVAE:
        def predict_on_batch(self, x):
            return x

    src = ['как определить тип личности по форме носа: метод аристотеля', 'какие преступления вдохновили достоевского',
           'майк тайсон - о пользе чтения']
    tok = DefaultTokenizer()
    bounds = [tok.tokenize_into_words(t) for t in src]
    texts = tuple(Conv1dTextVAE.tokenize(t, b) for t, b in zip(src, bounds))
    chars = tuple(tuple(tok.tokenize_into_characters(t, b)) for t, b in zip(src, bounds))
    alphabet = sorted(set(c for t in chars for c in t) | {'<BOS>', '<EOS>'})
    char_index = {c: i for i, c in enumerate(alphabet)}
    vocabulary, word_vectors = Conv1dTextVAE.prepare_vocabulary_and_word_vectors(texts, fasttext_model, None)
    L = max(len(c) for c in chars) + 3
    gen = SequenceForSeq2Seq(tokenizer=tok, input_texts=texts, target_texts=chars, batch_size=2, input_text_size=10,
                             output_text_size=L, input_vocabulary=vocabulary, input_word_vectors=word_vectors,
                             output_char_index=char_index, vae=FakeVAE())
    assert len(gen) == 1
    (words, gen_in), gen_out = gen[0]
    emb = fasttext_model.vector_size
    assert words.shape == (2, 10, emb + 1) and gen_in.shape == (2, L, len(alphabet)) and gen_out.shape == gen_in.shape
    for b in range(2):
        n = len(texts[b])
        # identity VAE: rows before the first end-of-sentence row survive, renormalised without the last slot
        for t in range(10):
            if t < min(n, 10):
                np.testing.assert_allclose(words[b, t], word_vectors[vocabulary[texts[b][t]], :emb + 1], atol=1e-5)
            else:
                assert not words[b, t].any()
        T = len(chars[b])
        assert gen_in[b, 0, char_index['<BOS>']] == 1.0
        for t in range(T):
            assert gen_in[b, t + 1, char_index[chars[b][t]]] == 1.0 and gen_out[b, t, char_index[chars[b][t]]] == 1.0
        assert gen_in[b, T + 1, char_index['<EOS>']] == 1.0
        assert gen_out[b, T, char_index['<EOS>']] == 1.0 and gen_out[b, T + 1, char_index['<EOS>']] == 1.0
        assert gen_in[b].sum() == T + 2 and gen_out[b].sum() == T + 2


def test_fit_predict_with_batch_norm(fasttext_model, corpus, tmp_path):
    """use_batch_norm=True (reference :148, :1229-1230, :1259-1260) end to end: the loss falls, the moving averages move
    and travel through pickling, and inference (which normalises with them) is reproduced bit for bit after unpickling."""
    from conv1d_text_vae_amd import Conv1dTextVAE
    vae = Conv1dTextVAE(fasttext_model, fasttext_model, max_epochs=8, batch_size=2, n_filters=(24, 16), latent_dim=12,
                        use_batch_norm=True)
    vae.fit(corpus[0], corpus[1])
    _assert_fitted(vae)
    hist = vae.vae_history_
    assert hist[-1]['loss'] < hist[0]['loss'] and np.isfinite(hist[-1]['val_loss'])
    w = vae._vae_runtime.weights()
    for name in ('encoder_batchnorm_1', 'encoder_batchnorm_2', 'deconv_decoder_batchnorm_1', 'deconv_decoder_batchnorm_2'):
        for part in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
            assert name + '/' + part in w
        assert np.abs(w[name + '/moving_mean']).max() > 0 and not np.allclose(w[name + '/moving_variance'], 1.0)
    assert any(k.startswith('encoder_batchnorm_1') for k in vae.vae_encoder_.get_weights())
    assert any(k.startswith('deconv_decoder_batchnorm_1') for k in vae.vae_decoder_.get_weights())
    out = vae.predict(corpus[0][:3])
    assert len(out) == 3 and all(isinstance(t, str) and len(t) > 0 for t in out)
    z1 = vae.transform(corpus[0])
    with open(str(tmp_path / 'bn.pkl'), 'wb') as fp:
        pickle.dump(vae, fp)
    with open(str(tmp_path / 'bn.pkl'), 'rb') as fp:
        res = pickle.load(fp)
    assert res.use_batch_norm is True
    res.input_embeddings = res.output_embeddings = fasttext_model
    np.testing.assert_array_equal(res.transform(corpus[0]), z1)
    assert res.predict(corpus[0][:3]) == out


def test_best_weights_checkpoint_carries_the_batchnorm_moving_averages(fasttext_model, corpus, monkeypatch):
    """ModelCheckpoint(save_weights_only=True, save_best_only=True) + load_weights (reference :312-316, :331-332) save and
    restore EVERY weight of the model, the moving mean / variance of the BatchNormalization layers included: after the
    best-epoch restore they must be those of the best epoch, not of the last one.  The fit loop is intercepted: snapshot,
    two more training steps (which move the averages), restore."""
    import torch
    from conv1d_text_vae_amd import Conv1dTextVAE
    from conv1d_text_vae_amd import estimator as est_mod
    seen = {}

    def probe(dp, B, n_tr, n_ev, max_epochs, train_slice, eval_slice, local_sums, get_lr, set_lr, snapshot, restore, **kw):
        eng = train_slice.__closure__ and [c.cell_contents for c in train_slice.__closure__
                                          if type(c.cell_contents).__name__ == 'VAEEngine'][0]
        if eng.bn and 'done' not in seen:
            first, count = dp.batch_slice(B)
            train_slice(first, count)
            snap = snapshot()
            ref = {n: (st['moving_mean'].clone(), st['moving_var'].clone()) for n, st in eng.bn.items()}
            p_ref = eng.params.clone()
            train_slice(first, count)
            train_slice(first, count)
            moved = any(not torch.equal(st['moving_mean'], ref[n][0]) for n, st in eng.bn.items())
            restore(snap)
            seen['done'] = moved and torch.equal(eng.params, p_ref) and all(
                torch.equal(st['moving_mean'], ref[n][0]) and torch.equal(st['moving_var'], ref[n][1]) for n, st in eng.bn.items())
            local_sums()
        return real(dp, B, n_tr, n_ev, max_epochs, train_slice, eval_slice, local_sums, get_lr, set_lr, snapshot, restore, **kw)
    real = est_mod.run_epochs_dp
    monkeypatch.setattr(est_mod, 'run_epochs_dp', probe)
    vae = Conv1dTextVAE(fasttext_model, fasttext_model, max_epochs=2, batch_size=2, n_filters=(16,), latent_dim=8, use_batch_norm=True)
    vae.fit(corpus[0], corpus[1])
    assert seen.get('done') is True
