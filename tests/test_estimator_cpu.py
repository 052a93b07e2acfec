"""Host logic of the drop-in estimator surface, CPU only.

These are the reference's own offline-runnable checks (tests/test_conv1d_text_vae.py: tokenizer :17-57, generator
invariants :60-267, helper and error-string tests :555-1112, :1327-1371, :1379-1413, :1431-1481) replayed against the
new package with a deterministic synthetic FastText provider (the reference downloads a real model in setUpClass; there
is no network here).  Nothing numerical of the hot path is exercised: that is tests/test_estimator_gpu.py.
"""
import copy
import csv
import os
import pickle
import re

import numpy as np
import pytest
from sklearn.exceptions import NotFittedError

from conv1d_text_vae_amd import (BaseTokenizer, Conv1dTextVAE, DefaultTokenizer, FastTextKeyedVectors,
                                 SequenceForVAE)
from conv1d_text_vae_amd import embeddings as E
from conv1d_text_vae_amd import training

HERE = os.path.dirname(os.path.abspath(__file__))


def load_testdata(max_target_words=None):
    """The reference's test corpus: 10 (annotation, full_text) pairs, lower-cased as tests:535-536 do.  `max_target_words`
    cuts the (very long) full texts so that the character-level stage of the GPU tests stays short."""
    inputs, targets = [], []
    with open(os.path.join(HERE, 'golden', 'testdata.csv'), 'r', encoding='utf-8', errors='ignore') as fp:
        header = None
        for row in csv.reader(fp, delimiter=',', quotechar='"'):
            if len(row) == 0:
                continue
            if header is None:
                header = row
                assert header == ['annotation', 'full_text']
            else:
                inputs.append(row[0].strip().lower())
                targets.append(row[1].strip().lower())
    if max_target_words is not None:
        targets = [' '.join(t.split()[:max_target_words]) for t in targets]
    return inputs, targets


@pytest.fixture(scope='module')
def corpus():
    return load_testdata()


@pytest.fixture(scope='module')
def fasttext_model(corpus):
    tok = DefaultTokenizer()
    words = []
    for t in corpus[0] + corpus[1]:
        words += list(Conv1dTextVAE.tokenize(t, tok.tokenize_into_words(t)))
    words += ['мама', 'мыла', 'раму', 'папа', 'мыл', 'синхрофазотрон', 'а', 'дочка', 'свой', 'ноутбук', 'от', 'коньяка',
              'нейросеть', 'мыть', 'рама']
    return E.synthetic_fasttext(words, vector_size=40, bucket=2000, seed=7)


@pytest.fixture()
def text_vae(fasttext_model):
    return Conv1dTextVAE(input_embeddings=fasttext_model, output_embeddings=fasttext_model, max_epochs=10, batch_size=2)


# ---------------------------------------------------------------------------------------------------------
# tokenizer (reference tests :17-57)
# ---------------------------------------------------------------------------------------------------------
def test_tokenize_into_words_positive01():
    assert DefaultTokenizer().tokenize_into_words(' Мама  мыла23\tраму.') == [(1, 5), (7, 11), (11, 13), (14, 18), (18, 19)]


def test_tokenize_into_words_positive02():
    tok = DefaultTokenizer(special_symbols={'\\n'})
    assert tok.tokenize_into_words(' Мама  мыла23\\nраму.') == [(1, 5), (7, 11), (11, 13), (13, 15), (15, 19), (19, 20)]


def test_tokenize_into_characters():
    s = ' Мама  мыла23 \n\tраму.'
    bounds = [(1, 5), (7, 11), (11, 13), (16, 20), (20, 21)]
    assert DefaultTokenizer().tokenize_into_characters(s, bounds) == \
        [' ', 'М', 'а', 'м', 'а', ' ', 'м', 'ы', 'л', 'а', '2', '3', '\n', 'р', 'а', 'м', 'у', '.']


def test_tokenizer_serialize():
    tok = DefaultTokenizer()
    res = pickle.loads(pickle.dumps(tok))
    assert isinstance(res, DefaultTokenizer) and res is not tok
    s = ' Мама  мыла23\tраму.'
    assert tok.tokenize_into_words(s) == res.tokenize_into_words(s)
    assert isinstance(copy.deepcopy(tok), DefaultTokenizer)


def test_tokenizer_punctuation_between_digits_and_symbols():
    tok = DefaultTokenizer()
    s = 'цена 3,5$ (примерно) - это 1.000.000 руб.'
    words = [s[a:b] for a, b in tok.tokenize_into_words(s)]
    assert words == ['цена', '3,5', '$', '(', 'примерно', ')', '-', 'это', '1.000.000', 'руб', '.']
    assert tok.tokenize_into_words('   ') == []


def test_tokenize():
    tok = DefaultTokenizer(special_symbols={'\n'})
    s = ' Мама  мыла23\nраму.'
    res = Conv1dTextVAE.tokenize(s, tok.tokenize_into_words(s))
    assert isinstance(res, tuple) and res == ('мама', 'мыла', '23', '\n', 'раму', '.')


# ---------------------------------------------------------------------------------------------------------
# embedding container
# ---------------------------------------------------------------------------------------------------------
def test_embedding_container_contract(fasttext_model, tmp_path):
    kv = fasttext_model
    assert kv.vector_size == 40
    v = kv['мама']
    assert v.shape == (40,) and v.dtype == np.float32
    oov = kv['несуществующееслово']                      # sub-word fallback, as FastText
    assert oov.shape == (40,) and np.linalg.norm(oov) > 0
    plain = FastTextKeyedVectors(vector_size=4, min_n=1, max_n=5)
    plain.add(['a', 'b'], np.eye(2, 4, dtype=np.float32))
    with pytest.raises(KeyError):
        _ = plain['zzz']
    name = str(tmp_path / 'kv.bin')
    kv.save(name)
    kv2 = FastTextKeyedVectors.load(name)
    assert kv2.index2word == kv.index2word
    np.testing.assert_array_equal(kv2['мама'], kv['мама'])
    np.testing.assert_array_equal(kv2['несуществующееслово'], oov)
    kv3 = Conv1dTextVAE.copy_embeddings(kv)
    assert kv3 is not kv and kv3.index2word == kv.index2word
    rows = kv.vectors_norm
    np.testing.assert_allclose(np.linalg.norm(rows, axis=1), 1.0, atol=1e-5)


def test_ft_hash_known_values():
    # FNV-1a 32-bit test vectors (ASCII): the FastText hash is plain FNV-1a on bytes < 128
    assert E.ft_hash('') == 0x811C9DC5
    assert E.ft_hash('a') == 0xE40C292C
    assert E.ft_hash('foobar') == 0xBF9CF968
    assert E.compute_ngrams('ab', 3, 4) == ['<ab', 'ab>', '<ab>']


def test_load_word2vec_text(tmp_path):
    p = tmp_path / 'tiny.vec'
    p.write_text('2 3\nfoo 1 0 0\nbar 0 2 0\n', encoding='utf-8')
    kv = E.load_word2vec_text(str(p))
    assert kv.vector_size == 3 and kv.index2word == ['foo', 'bar']
    np.testing.assert_array_equal(kv['bar'], np.array([0, 2, 0], dtype=np.float32))


# ---------------------------------------------------------------------------------------------------------
# estimator shell
# ---------------------------------------------------------------------------------------------------------
HYPER = ['n_filters', 'kernel_size', 'input_embeddings', 'output_embeddings', 'batch_size', 'max_epochs', 'lr',
         'latent_dim', 'n_recurrent_units', 'use_batch_norm', 'warm_start', 'verbose', 'use_attention', 'input_text_size',
         'output_text_size', 'validation_fraction', 'tokenizer']
FITTED = ['input_text_size_', 'output_text_size_', 'input_vector_size_', 'output_vector_size_', 'vae_encoder_',
          'vae_decoder_', 'generator_encoder_', 'generator_decoder_', 'output_text_size_in_characters_',
          'target_char_index_', 'reverse_target_char_index_']


def test_create(text_vae):
    assert isinstance(text_vae, Conv1dTextVAE)
    for name in HYPER:
        assert hasattr(text_vae, name), name
    for name in FITTED:
        assert not hasattr(text_vae, name), name
    assert text_vae.n_filters == 128 and text_vae.kernel_size == 3 and text_vae.latent_dim == 5
    assert text_vae.n_recurrent_units == 128 and text_vae.lr == 0.001 and text_vae.validation_fraction == 0.2


def test_prepare_vocabulary_and_word_vectors(fasttext_model):
    eps = 1e-5
    input_texts = (('мама', 'мыла', 'раму'), ('папа', 'мыл', 'синхрофазотрон', '!'),
                   ('а', 'дочка', 'мыла', 'свой', 'ноутбук', 'от', 'коньяка'))
    true_words = {'мама', 'мыла', 'раму', 'папа', 'мыл', 'синхрофазотрон', '!', 'а', 'дочка', 'свой', 'ноутбук', 'от',
                  'коньяка', ''}
    true_vector_size = fasttext_model.vector_size + 3
    # as in the reference's run, ONE corpus word has no vector (there: a word whose n-grams the real model lacks; here: a
    # container without sub-word fallback that does not know it), so it shares the unknown-word row
    known = [w for w in sorted(true_words - {'', '!', 'синхрофазотрон'})]
    model = FastTextKeyedVectors(vector_size=fasttext_model.vector_size, min_n=1, max_n=5)
    model.add(known, np.vstack([fasttext_model[w] for w in known]))
    vocabulary, word_vectors = Conv1dTextVAE.prepare_vocabulary_and_word_vectors(input_texts, model, ('!',))
    assert isinstance(vocabulary, dict) and set(vocabulary.keys()) == true_words
    assert vocabulary['синхрофазотрон'] == len(known)
    assert vocabulary[''] == len(true_words) - 1
    for w in true_words - {''}:
        assert 0 <= vocabulary[w] < len(true_words) - 1
    assert isinstance(word_vectors, np.ndarray) and word_vectors.shape == (len(true_words), true_vector_size)
    special = {len(true_words) - 1, vocabulary['!']}
    for i in range(word_vectors.shape[0]):
        assert abs(np.linalg.norm(word_vectors[i]) - 1.0) < 1e-4
        if i not in special:
            assert word_vectors[i, true_vector_size - 1] == 0.0 and word_vectors[i, true_vector_size - 2] == 0.0
            assert abs(word_vectors[i, true_vector_size - 3]) < eps or abs(word_vectors[i, true_vector_size - 3] - 1.0) < eps
    assert word_vectors[len(true_words) - 1, true_vector_size - 1] == 1.0
    assert word_vectors[vocabulary['!'], true_vector_size - 2] == 1.0


def test_get_vocabulary_and_word_vectors_from_fasttext(fasttext_model):
    input_texts = (('мама', 'мыла', 'раму'), ('папа', 'мыл', 'синхрофазотрон', '!'),
                   ('а', 'дочка', 'мыла', 'свой', 'ноутбук', 'от', 'коньяка'))
    true_words = {'мама', 'мыла', 'раму', 'папа', 'мыл', 'синхрофазотрон', '!', 'а', 'дочка', 'свой', 'ноутбук', 'от',
                  'коньяка', ''}
    vocabulary, word_vectors = Conv1dTextVAE.get_vocabulary_and_word_vectors_from_fasttext(
        input_texts, fasttext_model, ('мама',))
    assert set(vocabulary.keys()) < true_words and 'мама' not in vocabulary
    assert word_vectors.shape == (len(vocabulary), fasttext_model.vector_size)
    assert sorted(vocabulary.values()) == list(range(len(vocabulary)))
    np.testing.assert_allclose(np.linalg.norm(word_vectors, axis=1), 1.0, atol=1e-4)
    # first-occurrence order (:1015-1028)
    assert vocabulary['мыла'] == 0 and vocabulary['раму'] == 1 and vocabulary['папа'] == 2


def test_check_texts_param_negative(corpus):
    msg = re.escape('The parameter `X` is wrong! Expected `{0}`, `{1}` or 1-D `{2}`, got `{3}`.'.format(
        type([1, 2]), type((1, 2)), type(np.array([1, 2])), type({1, 2})))
    with pytest.raises(ValueError, match=msg):
        Conv1dTextVAE.check_texts_param(set(corpus[0]), 'X')
    texts = np.array([[1, 2, 3], [4, 5, 6]])
    with pytest.raises(ValueError, match=re.escape('The parameter `y` is wrong! Expected 1-D array, got 2-D array.')):
        Conv1dTextVAE.check_texts_param(texts, 'y')
    bad = copy.deepcopy(corpus[0])
    bad[3] = 4.5
    with pytest.raises(ValueError, match=re.escape('Item 3 of the parameter `X` is wrong! This item is not string!')):
        Conv1dTextVAE.check_texts_param(bad, 'X')


def _not_defined(name):
    return 'The parameter `{0}` is not defined!'.format(name)


def _wrong_type(name, expected, got):
    return 'The parameter `{0}` is wrong! Expected `{1}`, got `{2}`.'.format(name, type(expected), type(got))


def _not_positive(name, v):
    return 'The parameter `{0}` is wrong! Expected a positive value, but {1} is not positive.'.format(name, v)


CHECK_PARAMS_CASES = (
    [(n, 'del', _not_defined(n)) for n in ('warm_start', 'use_batch_norm', 'use_attention', 'verbose', 'batch_size',
                                           'max_epochs', 'latent_dim', 'input_text_size', 'output_text_size',
                                           'output_onehot_size', 'n_filters', 'kernel_size', 'n_recurrent_units',
                                           'validation_fraction', 'lr')] +
    [(n, 0.5, _wrong_type(n, True, 0.5)) for n in ('warm_start', 'use_batch_norm', 'use_attention', 'verbose')] +
    [(n, 4.5, _wrong_type(n, 10, 4.5)) for n in ('batch_size', 'max_epochs', 'latent_dim', 'input_text_size',
                                                 'output_text_size', 'output_onehot_size', 'kernel_size',
                                                 'n_recurrent_units')] +
    [(n, -3, _not_positive(n, -3)) for n in ('batch_size', 'max_epochs', 'latent_dim', 'input_text_size',
                                             'output_text_size', 'output_onehot_size', 'kernel_size', 'n_recurrent_units',
                                             'n_filters')] +
    [('n_filters', 4.5, 'The parameter `n_filters` is wrong! Expected `{0}` or `{1}`, got `{2}`.'.format(
        type(10), type((1, 2)), type(4.5))),
     ('n_filters', tuple(), 'The parameter `n_filters` is wrong! Expected a nonempty sequence of integers.'),
     ('n_filters', (128, 0, 64), 'Item 1 of the parameter `n_filters` is wrong! Expected a positive value, but 0 is not '
                                 'positive.'),
     ('validation_fraction', 2, _wrong_type('validation_fraction', 10.5, 2)),
     ('validation_fraction', 1.5, 'The parameter `validation_fraction` is wrong! Expected a positive value between 0.0 and '
                                  '1.0, but 1.5 does not correspond to this condition.'),
     ('validation_fraction', 0.0, 'The parameter `validation_fraction` is wrong! Expected a positive value between 0.0 and '
                                  '1.0, but 0 does not correspond to this condition.'),
     ('lr', 1, _wrong_type('lr', 10.5, 1)),
     ('lr', -0.5, _not_positive('lr', -0.5))]
)


@pytest.mark.parametrize('name,value,message', CHECK_PARAMS_CASES)
def test_check_params_negative(text_vae, name, value, message):
    params = dict(text_vae.__dict__)
    if isinstance(value, str) and value == 'del':
        del params[name]
    else:
        params[name] = value
    with pytest.raises(ValueError, match=re.escape(message)):
        Conv1dTextVAE.check_params(**params)


def test_check_params_positive(text_vae):
    Conv1dTextVAE.check_params(**text_vae.get_params(deep=False))
    params = text_vae.get_params(deep=True)
    assert set(params) == set(HYPER + ['output_onehot_size'])
    text_vae.set_params(**dict(params, latent_dim=7))
    assert text_vae.latent_dim == 7


def test_float_to_string():
    assert Conv1dTextVAE.float_to_string(3.567) == '3.567'
    assert Conv1dTextVAE.float_to_string(0.50670) == '0.5067'
    assert Conv1dTextVAE.float_to_string(12) == '12'


@pytest.mark.parametrize('special_symbols', [None, ('\t', '\n')])
def test_texts_to_data(corpus, fasttext_model, special_symbols):
    """reference tests :999-1112"""
    batch_size, input_text_size, n_texts = 2, 10, 3
    tokenizer = DefaultTokenizer()
    texts = corpus[0][:n_texts]
    data = list(Conv1dTextVAE.texts_to_data(texts, batch_size=batch_size, max_text_size=input_text_size,
                                            tokenizer=tokenizer, fasttext_model=fasttext_model,
                                            special_symbols=special_symbols))
    assert len(data) == 2
    V = fasttext_model.vector_size + 2 + (0 if special_symbols is None else len(special_symbols))
    text_idx = 0
    for batch in data:
        assert isinstance(batch, np.ndarray) and batch.shape == (batch_size, input_text_size, V) and batch.dtype == np.float32
        for sample in range(batch_size):
            n_tokens = len(Conv1dTextVAE.tokenize(texts[text_idx], tokenizer.tokenize_into_words(texts[text_idx])))
            for t in range(input_text_size):
                norm = np.linalg.norm(batch[sample][t])
                if t <= n_tokens:
                    assert abs(norm - 1.0) < 1e-4
                    assert batch[sample][t][V - 1] == (0.0 if t < n_tokens else 1.0)
                else:
                    assert abs(norm) < 1e-4
            if text_idx < n_texts - 1:
                text_idx += 1


def test_texts_to_data_special_and_unknown_slots():
    kv = FastTextKeyedVectors(vector_size=4, min_n=1, max_n=5)          # no sub-words: unknown words raise
    kv.add(['foo'], np.array([[0, 3, 0, 4]], dtype=np.float32))
    tok = DefaultTokenizer(special_symbols={'@@'})
    data = next(Conv1dTextVAE.texts_to_data(['foo @@ bar'], 1, 5, tok, kv, ('@@',)))
    assert data.shape == (1, 5, 7)
    np.testing.assert_allclose(data[0, 0], [0, 0.6, 0, 0.8, 0, 0, 0])
    np.testing.assert_array_equal(data[0, 1], [0, 0, 0, 0, 1, 0, 0])    # special symbol i -> emb + i
    np.testing.assert_array_equal(data[0, 2], [0, 0, 0, 0, 0, 1, 0])    # unknown -> V - 2
    np.testing.assert_array_equal(data[0, 3], [0, 0, 0, 0, 0, 0, 1])    # end of sentence -> V - 1
    assert not data[0, 4].any()


@pytest.mark.parametrize('special_symbols', [None, ('\t', '\n')])
def test_sequence_for_vae(fasttext_model, special_symbols):
    """reference tests :71-267"""
    src = ['как определить тип личности\tпо форме носа:\nметод аристотеля' if special_symbols else
           'как определить тип личности по форме носа: метод аристотеля',
           'какие преступления вдохновили достоевского', 'майк тайсон - о пользе чтения']
    batch_size, T_in, T_out = 2, 10, 9
    tokenizer = DefaultTokenizer(special_symbols=set(special_symbols) if special_symbols else None)
    texts = tuple(Conv1dTextVAE.tokenize(t, tokenizer.tokenize_into_words(t)) for t in src)
    vocabulary, word_vectors = Conv1dTextVAE.prepare_vocabulary_and_word_vectors(texts, fasttext_model, special_symbols)
    gen = SequenceForVAE(tokenizer=tokenizer, input_texts=texts, target_texts=texts, batch_size=batch_size,
                         input_text_size=T_in, output_text_size=T_out, input_vocabulary=vocabulary,
                         output_vocabulary=vocabulary, input_word_vectors=word_vectors, output_word_vectors=word_vectors)
    assert len(gen) == 1
    n_special = 0 if special_symbols is None else len(special_symbols)
    emb = fasttext_model.vector_size
    x, y = gen[0]
    assert isinstance(gen[0], tuple)
    assert x.shape == (batch_size, T_in, emb + 2 + n_special) and y.shape == (batch_size, T_out, word_vectors.shape[0])
    in_ids, out_ids = gen.ids(0)
    for sample in range(batch_size):
        tokens = texts[sample]
        n_tokens = len(tokens)
        for t in range(T_in):
            norm = np.linalg.norm(x[sample][t])
            if t <= n_tokens:
                assert abs(norm - 1.0) < 1e-4
                assert x[sample][t][emb + 1 + n_special] == (0.0 if t < n_tokens else 1.0)
                if t < n_tokens and special_symbols and tokens[t] in special_symbols:
                    assert x[sample][t][emb + 1 + special_symbols.index(tokens[t])] == 1.0
            else:
                assert norm == 0.0 and in_ids[sample, t] == -1
        for t in range(T_out):
            if t <= n_tokens:
                assert abs(np.linalg.norm(y[sample][t]) - 1.0) < 1e-6
                best = int(np.argmax(y[sample][t]))
                assert y[sample][t][best] == 1.0 and out_ids[sample, t] == best
                if t < n_tokens:
                    if special_symbols and tokens[t] in special_symbols:
                        assert best == vocabulary[tokens[t]]
                    else:
                        assert best < vocabulary[''] - n_special
                else:
                    assert best == vocabulary['']
            else:
                assert np.linalg.norm(y[sample][t]) == 0.0 and out_ids[sample, t] == -1


def test_fit_negative(text_vae, corpus):
    X, y = corpus
    with pytest.raises(ValueError, match=re.escape('Length of `X` does not equal to length of `y`! {0} != {1}.'.format(
            len(X), len(y) - 1))):
        text_vae.fit(X, y[:-1])
    bad = copy.deepcopy(X)
    bad[1] = 2
    with pytest.raises(ValueError, match=re.escape('Item 1 of the parameter `X` is wrong! This item is not string!')):
        text_vae.fit(bad, y)
    text_vae.batch_size = -3
    with pytest.raises(ValueError, match=re.escape(_not_positive('batch_size', -3))):
        text_vae.fit(X, y)
    text_vae.batch_size = 2
    emb = text_vae.input_embeddings
    del text_vae.input_embeddings
    with pytest.raises(ValueError, match=re.escape('The parameter `input_embeddings` is not defined!')):
        text_vae.fit(X, y)
    text_vae.input_embeddings = 4
    with pytest.raises(ValueError, match=re.escape('The parameter `input_embeddings` is wrong! Expected `{0}`, got `{1}`.'.format(
            type(FastTextKeyedVectors(vector_size=300, min_n=1, max_n=5)), type(4)))):
        text_vae.fit(X, y)
    text_vae.input_embeddings = emb
    del text_vae.output_embeddings
    with pytest.raises(ValueError, match=re.escape('The parameter `output_embeddings` is not defined!')):
        text_vae.fit(X, y)
    text_vae.output_embeddings = 3.5
    with pytest.raises(ValueError, match=re.escape('The parameter `output_embeddings` is wrong! Expected `{0}`, got `{1}`.'.format(
            type(FastTextKeyedVectors(vector_size=300, min_n=1, max_n=5)), type(3.5)))):
        text_vae.fit(X, y)
    text_vae.output_embeddings = emb
    text_vae.validation_fraction = 0.01
    with pytest.raises(ValueError, match=re.escape('`validation_fraction` is too small! There are no samples for evaluation!')):
        text_vae.fit(X, y)
    text_vae.validation_fraction = 0.99
    with pytest.raises(ValueError, match=re.escape('`validation_fraction` is too large! There are no samples for training!')):
        text_vae.fit(X, y)


def test_not_fitted(text_vae, corpus):
    with pytest.raises(NotFittedError):
        text_vae.transform(corpus[0])
    with pytest.raises(NotFittedError):
        text_vae.predict(corpus[0])


def test_serialize_unfitted(text_vae):
    res = pickle.loads(pickle.dumps(text_vae))
    assert isinstance(res, Conv1dTextVAE) and res is not text_vae
    for name in HYPER:
        assert hasattr(res, name)
    for name in FITTED:
        assert not hasattr(res, name)
    assert res.input_embeddings is None and res.output_embeddings is None
    for name in ('n_filters', 'kernel_size', 'batch_size', 'use_attention', 'use_batch_norm', 'max_epochs', 'latent_dim',
                 'n_recurrent_units', 'warm_start', 'verbose', 'input_text_size', 'output_text_size',
                 'validation_fraction', 'lr'):
        assert getattr(res, name) == getattr(text_vae, name)


def test_setstate_validation(text_vae):
    state = text_vae.__getstate__()
    with pytest.raises(ValueError, match=re.escape('`state` is wrong! Expected {0}.'.format(type({0: 1})))):
        Conv1dTextVAE.__new__(Conv1dTextVAE).__setstate__([1, 2])
    fitted = dict(state, input_text_size_=5, output_text_size_=5, input_vector_size_=42, output_vector_size_=42,
                  output_text_size_in_characters_=9, target_char_index_={'a': 0}, reverse_target_char_index_={0: 'a'},
                  weights_=3)
    with pytest.raises(ValueError, match=re.escape('Weights are wrong! Expected `{0}`, got `{1}`.'.format(
            type({'a': 1}), type(3)))):
        Conv1dTextVAE.__new__(Conv1dTextVAE).__setstate__(dict(fitted))
    fitted['weights_'] = {'vae_encoder': b''}
    with pytest.raises(ValueError, match=re.escape('Weights are wrong! The key `vae_decoder` is not found.')):
        Conv1dTextVAE.__new__(Conv1dTextVAE).__setstate__(dict(fitted))


def test_find_best_texts():
    variants = [(('мама', 0.0), ('папа', 0.3)), (('мыть', 0.0), ('мыла', 0.1), ('мыл', 0.5)), (('рама', 0.0),)]
    res = Conv1dTextVAE.find_best_texts(variants, 3)
    assert res == ['мама мыть рама', 'мама мыла рама', 'папа мыла рама']
    assert len(Conv1dTextVAE.find_best_texts(variants, 10)) == 4


def test_postprocess_negative():
    with pytest.raises(ValueError, match=re.escape('The `word_vectors` parameter is wrong! Expected 3-D array, got 2-D array!')):
        Conv1dTextVAE.postprocess_reconstructed_word_vectors(np.zeros((20, 7)))


# ---------------------------------------------------------------------------------------------------------
# fit-loop policy (Keras callbacks restated)
# ---------------------------------------------------------------------------------------------------------
def test_callbacks_follow_keras_rules():
    stop = training.EarlyStopping(patience=2)
    for v in (3.0, 2.0, 2.0, 2.5):
        stop.on_epoch_end(v)
    assert stop.stop and stop.best == 2.0
    lr = [0.002]
    plateau = training.ReduceLROnPlateau(lambda: lr[0], lambda v: lr.__setitem__(0, v), factor=0.1, patience=2)
    for v in (1.0, 0.99995, 0.99991):        # improvements below min_delta=1e-4 do not count
        plateau.on_epoch_end(v)
    assert abs(lr[0] - 0.0002) < 1e-12 and plateau.wait == 0
    best = training.BestWeights(lambda: 'w%d' % len(seen))
    seen = []
    for v in (2.0, 1.0, 1.5):
        seen.append(v)
        best.on_epoch_end(v)
    assert best.weights == 'w2' and best.best == 1.0


def test_run_epochs_shuffles_batches_and_restores_best():
    log, val = [], iter([5.0, 4.0, 4.5, 4.6, 4.7, 4.8, 4.9, 5.0])
    state = {'w': 0, 'lr': 0.002, 'cur': None}

    def train_batch(i):
        log.append(i)
        state['w'] += 1

    def epoch_losses():
        state['cur'] = next(val)
        return 0.0, state['cur']

    hist = training.run_epochs(4, 1, 8, train_batch, lambda i: None, epoch_losses, lambda: state['lr'],
                               lambda v: state.__setitem__('lr', v), lambda: state['w'],
                               lambda w: state.__setitem__('w', w), rng=np.random.RandomState(0))
    assert len(hist) == 7                                   # best at epoch 2, patience 5 -> stops after epoch 7
    assert state['w'] == 8                                  # weights of epoch 2 restored (2 epochs x 4 batches)
    assert sorted(log[:4]) == [0, 1, 2, 3] and any(log[4 * e:4 * e + 4] != [0, 1, 2, 3] for e in range(7))
    assert state['lr'] < 0.002
    with pytest.raises(ValueError):
        training.run_epochs(0, 1, 1, train_batch, train_batch, epoch_losses, None, None, None, None)


# ---- FastText .bin reader (reference fasttext_loading.py:27, :43, :59) ---------------------------------
def _write_fasttext_bin(path, words, dim, minn, maxn, bucket, rng, labels=()):
    """A tiny native FastText model written field by field (the published layout of the format)."""
    import struct
    entries = [(w, 0) for w in words] + [(lb, 1) for lb in labels]
    mat = rng.standard_normal((len(words) + bucket, dim)).astype('<f4')
    with open(path, 'wb') as fp:
        fp.write(struct.pack('<ii', 793712314, 12))
        fp.write(struct.pack('<12i', dim, 5, 5, 1, 5, 1, 1, 2, bucket, minn, maxn, 100))
        fp.write(struct.pack('<d', 1e-4))
        fp.write(struct.pack('<3i', len(entries), len(words), len(labels)))
        fp.write(struct.pack('<qq', 12345, -1))
        for i, (w, kind) in enumerate(entries):
            fp.write(w.encode('utf-8') + b'\x00')
            fp.write(struct.pack('<qb', 100 - i, kind))
        fp.write(struct.pack('<?', False))
        fp.write(struct.pack('<qq', mat.shape[0], dim))
        fp.write(mat.tobytes())
        fp.write(struct.pack('<?', False))                   # output matrix: present in real files, never read
        fp.write(struct.pack('<qq', len(words), dim))
        fp.write(np.zeros((len(words), dim), dtype='<f4').tobytes())
    return mat


def test_fasttext_bin_reader_round_trip(tmp_path):
    from conv1d_text_vae_amd import embeddings as E
    rng = np.random.RandomState(0)
    words = ['мама', 'мыла', 'раму', 'the', 'cat', '</s>']
    dim, minn, maxn, bucket = 12, 3, 5, 50
    path = str(tmp_path / 'tiny.bin')
    mat = _write_fasttext_bin(path, words, dim, minn, maxn, bucket, rng, labels=('__label__x',))
    kv = E.load_fasttext_bin(path)
    assert isinstance(kv, E.KEYED_VECTOR_TYPES) and kv.vector_size == dim and kv.index2word == words
    assert (kv.min_n, kv.max_n, kv.bucket) == (minn, maxn, bucket)
    np.testing.assert_array_equal(kv.vectors_ngrams, mat[len(words):])
    for i, w in enumerate(words):
        grams = E.compute_ngrams(w, minn, maxn)
        want = (mat[i] + sum(mat[len(words) + E.ft_hash(g) % bucket] for g in grams)) / (len(grams) + 1)
        np.testing.assert_allclose(kv[w], want, rtol=1e-6, atol=1e-7)
    # an out-of-vocabulary word: the mean of its n-gram rows (FastText's sub-word fallback)
    grams = E.compute_ngrams('мамонт', minn, maxn)
    want = sum(mat[len(words) + E.ft_hash(g) % bucket] for g in grams) / len(grams)
    np.testing.assert_allclose(kv['мамонт'], want, rtol=1e-6, atol=1e-7)
    # `limit` keeps the most frequent words and still reads the n-gram rows behind ALL words
    kv2 = E.load_fasttext_bin(path, limit=2)
    assert kv2.index2word == words[:2]
    np.testing.assert_array_equal(kv2.vectors_ngrams, mat[len(words):])
    np.testing.assert_allclose(kv2['мама'], kv['мама'])
    # the estimator accepts the object as its embedding model
    from conv1d_text_vae_amd import Conv1dTextVAE
    assert Conv1dTextVAE.calc_vector_size(kv, None) == dim + 2
    Conv1dTextVAE(input_embeddings=kv, output_embeddings=kv).check_embeddings()


def test_fasttext_bin_reader_refuses_what_it_cannot_read(tmp_path):
    from conv1d_text_vae_amd import embeddings as E
    bad = tmp_path / 'bad.bin'
    bad.write_bytes(b'\x01\x02\x03\x04' * 40)
    with pytest.raises(ValueError, match='is not a FastText model'):
        E.load_fasttext_bin(str(bad))
    with pytest.raises(FileNotFoundError):
        E.load_fasttext_bin(str(tmp_path / 'absent.bin'))
    # a gzip of a model is opened in place
    import gzip
    rng = np.random.RandomState(1)
    path = str(tmp_path / 'm.bin')
    _write_fasttext_bin(path, ['a', 'b'], 4, 2, 3, 10, rng)
    with open(path, 'rb') as src, gzip.open(path + '.gz', 'wb') as dst:
        dst.write(src.read())
    assert E.load_fasttext_bin(path + '.gz').index2word == ['a', 'b']
    # truncated matrix
    data = open(path, 'rb').read()
    (tmp_path / 'cut.bin').write_bytes(data[:len(data) // 2])
    with pytest.raises(ValueError, match='truncated|ends inside'):
        E.load_fasttext_bin(str(tmp_path / 'cut.bin'))
