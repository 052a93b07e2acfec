"""GPU evidence for the SURVEY.md 8(f) "next" rows that were built but unpinned on the device:

  * row 2, device-side batch assembly: c1v_gather_input against the reference-contract generator
    SequenceForVAE.__getitem__ (reference :1374-1390) -- bit-exact in fp32, and a train step fed from word ids equals the
    step fed from the dense batch;
  * row 1, the character-level GRU generator on cuda:0 against oracle/seq2seq_oracle.py (state, loss, one
    RMSprop(clipnorm) step), and greedy decoding on the device against the reference's per-character loop.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import seq2seq_oracle as SO
from oracle import vae_oracle as O

pytestmark = pytest.mark.gpu


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ------------------------------------------------------------------------------------------------ f2: batch assembly
def _corpus(rng, n_texts, n_words, t_max):
    words = ['w%d' % i for i in range(n_words)]
    return tuple(tuple(words[j] for j in rng.randint(0, n_words, size=rng.randint(1, t_max + 3))) for _ in range(n_texts))


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_gather_input_equals_the_reference_batch_generator(dtype):
    from conv1d_text_vae_amd import _lib as L
    from conv1d_text_vae_amd import engine, geometry
    from conv1d_text_vae_amd.estimator import SequenceForVAE
    rng = np.random.RandomState(3)
    E, n_words, B, T_in, T_out = 50, 40, 16, 9, 8
    V = E + 2
    table = O.synthetic_vocab(n_words, E, seed=4)                     # [n_words + 2, V]: words, unknown row, EOS row
    vocab = {'w%d' % i: i for i in range(n_words)}
    vocab[''] = n_words + 1                                            # vocabulary[''] = end-of-sentence row (:1078)
    texts_in, texts_out = _corpus(rng, 3 * B, n_words, T_in), _corpus(rng, 3 * B, n_words, T_out)
    seq = SequenceForVAE(tokenizer=None, input_texts=texts_in, target_texts=texts_out, batch_size=B, input_text_size=T_in,
                         output_text_size=T_out, input_vocabulary=vocab, input_word_vectors=table, output_vocabulary=vocab,
                         output_word_vectors=table)
    g = geometry.Geometry(T_in, T_out, V, V, (12,), 3, 6)
    eng = engine.VAEEngine(g, B, table, dtype=dtype, seed=2)
    eng.set_input_table(table)
    lib = L.load()
    for idx in range(len(seq)):
        x_ref, y_ref = seq[idx]                                        # dense float32 input, dense one-hot target
        in_ids, out_ids = seq.ids(idx)
        # texts longer than the window have no EOS row; shorter ones: EOS row, then all-zero rows (-1)
        assert (in_ids >= -1).all() and ((in_ids == -1).sum(axis=1) == np.maximum(T_in - 1 - np.array(
            [len(t) for t in texts_in[idx * B:(idx + 1) * B]]), 0)).all()
        eng.in_ids.copy_(torch.from_numpy(in_ids))
        eng.Xe[0].t.fill_(7.0)                                         # the kernel must also write the zero rows / channels
        L.check(lib.c1v_gather_input(C.byref(eng._gather_args), _stream()), 'c1v_gather_input')
        got = eng.Xe[0].valid().float().cpu().numpy()
        want = x_ref if dtype == 'f32' else torch.from_numpy(x_ref).to(torch.bfloat16).float().numpy()
        assert np.array_equal(got, want)                               # bit-exact (bf16: the rounded table rows)
        pad = eng.Xe[0].t[:B * eng.Xe[0].bstride].view(B, eng.Xe[0].pitch, eng.Xe[0].cp)[:, :T_in, V:]
        assert float(pad.float().abs().max()) == 0.0                   # channel padding is zeroed
        # the one-hot target of the reference generator <-> the int32 ids the loss kernels take
        want_ids = np.where(y_ref.any(axis=2), y_ref.argmax(axis=2), -1)
        assert np.array_equal(out_ids, want_ids)
        if idx == 0 and eng.compact:
            # the one-launch form for large batches (live-row table + gather, 1024-thread blocks): the same bits as the two
            eng.load_batch(ids=out_ids)
            eng.ops_live_rows[0]()
            torch.cuda.synchronize()
            pos0, ctl0 = eng.row_pos.clone(), eng.live_ctl.clone()
            eng.row_pos.fill_(-7); eng.live_ctl.zero_(); eng.Xe[0].t.fill_(7.0)
            eng._live_gather_op()()
            torch.cuda.synchronize()
            assert torch.equal(eng.row_pos, pos0) and torch.equal(eng.live_ctl, ctl0)
            assert np.array_equal(eng.Xe[0].valid().float().cpu().numpy(), want)
        if idx == 0:
            # a table whose rows are NOT whole 16-byte words (ld = V + 1, odd): the element-wise path of the kernel
            tab2 = torch.zeros(table.shape[0], V + 1, dtype=torch.float32, device='cuda')
            tab2[:, :V] = torch.from_numpy(table)
            a2 = L.GatherInputArgs(tab2.data_ptr(), V + 1, eng.in_ids.data_ptr(), T_in, V, eng.Xe[0].seq())
            eng.Xe[0].t.fill_(7.0)
            L.check(lib.c1v_gather_input(C.byref(a2), _stream()), 'c1v_gather_input')
            assert np.array_equal(eng.Xe[0].valid().float().cpu().numpy(), want)
    # a train step fed from ids is the train step fed from the dense batch, to the bit
    x_ref, _ = seq[1]
    in_ids, out_ids = seq.ids(1)
    res = []
    for from_ids in (False, True):
        e2 = engine.VAEEngine(g, B, table, dtype=dtype, seed=2)
        e2.set_input_table(table)
        e2.load_batch(x=x_ref, ids=out_ids)
        e2.in_ids.copy_(torch.from_numpy(in_ids))
        e2.train_step(use_graph=False, from_ids=from_ids)
        torch.cuda.synchronize()
        res.append((e2.loss(), e2.params.cpu().numpy().copy()))
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1])


# ------------------------------------------------------------------------------------------------ f1: GRU generator
def _generator_case(use_attention, device, seed=0, B=5, T=11, F=14, Cn=17, U=12, L=13):
    from conv1d_text_vae_amd.seq2seq import Seq2SeqGenerator
    gen = Seq2SeqGenerator(text_size=T, feature_size=F, n_chars=Cn, units=U, use_attention=use_attention, lr=0.01,
                           device=device, seed=seed)
    rng = np.random.RandomState(seed)
    w = gen.get_weights()
    for k in w:
        if k.endswith('/bias'):
            w[k] = rng.uniform(-0.3, 0.3, w[k].shape).astype(np.float32)
    gen.set_weights(w)
    words = rng.standard_normal((B, T, F)).astype(np.float32)
    words[0, 6:] = 0.0
    words[1, 3] = 0.0
    words[2, :] = 0.0
    cin = np.zeros((B, L, Cn), dtype=np.float32)
    cout = np.zeros((B, L, Cn), dtype=np.float32)
    for b in range(B):
        n = rng.randint(2, L - 1)
        cin[b, np.arange(n), rng.randint(0, Cn, size=n)] = 1.0
        cout[b, np.arange(n), rng.randint(0, Cn, size=n)] = 1.0
    return gen, words, cin, cout


@pytest.mark.parametrize('use_attention', [True, False])
def test_generator_on_the_gpu_follows_the_keras_restatement(use_attention):
    gen, words, cin, cout = _generator_case(use_attention, 'cuda:0')
    dev = gen.dev
    assert dev.type == 'cuda'
    p64 = {k: v.astype(np.float64) for k, v in gen.get_weights().items()}
    ref_loss, ref_state, ref_prob = SO.seq2seq_loss(p64, words.astype(np.float64), cin.astype(np.float64),
                                                    cout.astype(np.float64), use_attention)
    tw, ti, to = (torch.from_numpy(a).to(dev) for a in (words, cin, cout))
    with torch.no_grad():
        state = gen.encode(tw).cpu().numpy()
        loss = float(gen.loss(tw, ti, to, train=False))
    np.testing.assert_allclose(state, ref_state, rtol=2e-5, atol=2e-6)
    assert np.abs(state[2]).max() == 0.0
    assert abs(loss - ref_loss) < 2e-5 * max(1.0, abs(ref_loss))
    probs, h = gen.decoder_model().predict_on_batch([cin[:, :1], ref_state.astype(np.float32)])
    _, h_ref = SO.gru(cin[:, :1].astype(np.float64), ref_state, p64['seq2seq_decoder_gru/kernel'],
                      p64['seq2seq_decoder_gru/recurrent_kernel'], p64['seq2seq_decoder_gru/bias'], False)
    np.testing.assert_allclose(h, h_ref, rtol=2e-5, atol=2e-6)
    assert np.allclose(probs.sum(-1), 1.0, atol=1e-5)


def test_generator_gradient_and_rmsprop_step_on_the_gpu_follow_keras():
    """(1) The autograd gradient of the (test-phase) loss on the device against central differences of the oracle's loss;
    (2) one RMSprop(lr, rho 0.9, eps 1e-7, clipnorm 10) step: the training-phase gradient (dropout masks replayed from the
    generator's device RNG state) pushed through Keras' update rule in NumPy float64."""
    gen, words, cin, cout = _generator_case(True, 'cuda:0', seed=3)
    dev = gen.dev
    tw, ti, to = (torch.from_numpy(a).to(dev) for a in (words, cin, cout))
    before = {k: v.copy() for k, v in gen.get_weights().items()}
    p64 = {k: v.astype(np.float64) for k, v in before.items()}
    w64, i64, o64 = words.astype(np.float64), cin.astype(np.float64), cout.astype(np.float64)
    for v in gen.p.values():
        v.grad = None
    gen.loss(tw, ti, to, train=False).backward()
    auto = {k: v.grad.detach().cpu().numpy().astype(np.float64) for k, v in gen.p.items()}
    rng = np.random.RandomState(0)
    for k in sorted(p64):
        for _ in range(3):
            idx = tuple(rng.randint(0, s) for s in p64[k].shape)
            h = 1e-5
            up, dn = {kk: vv.copy() for kk, vv in p64.items()}, {kk: vv.copy() for kk, vv in p64.items()}
            up[k][idx] += h
            dn[k][idx] -= h
            num = (SO.seq2seq_loss(up, w64, i64, o64, True)[0] - SO.seq2seq_loss(dn, w64, i64, o64, True)[0]) / (2 * h)
            assert abs(auto[k][idx] - num) < 2e-3 * max(abs(num), 1e-2), (k, idx, auto[k][idx], num)
    st = gen.gen.get_state()
    for v in gen.p.values():
        v.grad = None
    loss = gen.loss(tw, ti, to, train=True)
    loss.backward()
    grads = {k: v.grad.detach().cpu().numpy().astype(np.float64).copy() for k, v in gen.p.items()}
    gen.gen.set_state(st)
    got_loss = float(gen.train_batch(tw, ti, to))
    assert abs(got_loss - float(loss)) < 1e-6
    norm = np.sqrt(sum((g * g).sum() for g in grads.values()))
    scale = 10.0 / norm if norm >= 10.0 else 1.0
    after = gen.get_weights()
    for k, g in grads.items():
        g = g * scale
        want = before[k] - 0.01 * g / (np.sqrt(0.1 * g * g) + 1e-7)
        np.testing.assert_allclose(after[k], want, rtol=2e-4, atol=2e-6)


def test_greedy_decode_on_the_gpu_equals_the_reference_character_loop():
    gen, words, _, _ = _generator_case(True, 'cuda:0', seed=5)
    with torch.no_grad():
        state = gen.encode(torch.from_numpy(words).to(gen.dev))
    bos, eos, max_chars = 0, gen.C - 1, 9
    got = gen.greedy_decode(state.clone(), bos, eos, max_chars)
    dec = gen.decoder_model()
    st = state.cpu().numpy()
    B = st.shape[0]
    seq = np.zeros((B, 1, gen.C), dtype=np.float32)
    seq[:, 0, bos] = 1.0
    stop, out = [False] * B, [[] for _ in range(B)]
    while not all(stop):                                                # reference :455-468
        probs, st = dec.predict_on_batch([seq, st])
        idx = np.argmax(probs[:, -1, :], axis=1)
        for b in range(B):
            if stop[b]:
                continue
            out[b].append(int(idx[b]))
            if idx[b] == eos or len(out[b]) > max_chars:
                stop[b] = True
            seq[b, 0, :] = 0.0
            seq[b, 0, idx[b]] = 1.0
    for b in range(B):
        row = got[b]
        hits = np.nonzero(row == eos)[0]
        n = int(hits[0]) + 1 if hits.size else row.shape[0]
        assert list(row[:n]) == out[b]


# ------------------------------------------------------------------------------------------------ f4: vocabulary quantiser
def _planted_vectors(rng, n_clusters=25, per=16, n_lonely=30, dim=48, spread=0.03):
    centres = rng.standard_normal((n_clusters, dim))
    centres /= np.linalg.norm(centres, axis=1, keepdims=True)
    rows, owner = [], []
    for c in range(n_clusters):
        rows.append(centres[c] + spread * rng.standard_normal((per, dim)))
        owner += [c] * per
    lonely = rng.standard_normal((n_lonely, dim))
    rows.append(lonely)
    owner += list(range(n_clusters, n_clusters + n_lonely))
    v = np.concatenate(rows).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    perm = rng.permutation(v.shape[0])
    return v[perm], np.asarray(owner)[perm]


def test_quantize_word_vectors_matches_an_exact_knn_dbscan_restatement():
    """`output_onehot_size` (reference :916-1007): the GPU neighbour search + scikit-learn DBSCAN against the same
    pipeline with the neighbours taken from a float64 NumPy distance matrix; planted clusters come out as clusters."""
    from scipy.sparse import csr_matrix
    from sklearn.cluster import DBSCAN
    from conv1d_text_vae_amd import Conv1dTextVAE
    rng = np.random.RandomState(11)
    v, owner = _planted_vectors(rng)
    n = v.shape[0]
    max_vocab = 60
    clusters, centers = Conv1dTextVAE.quantize_word_vectors(v, max_vocab, verbose=False)
    assert isinstance(clusters, list) and len(clusters) == n and centers.dtype == np.float32
    assert centers.shape == (max(clusters) + 1, v.shape[1])
    np.testing.assert_allclose(np.linalg.norm(centers, axis=1), 1.0, atol=1e-5)
    # restatement: exact neighbours from float64 cosines
    k = min(3 * max(1, n // max_vocab), n - 1)
    cos = v.astype(np.float64) @ v.astype(np.float64).T
    np.fill_diagonal(cos, 2.0)                                           # the word itself first
    nb = np.argsort(-cos, axis=1, kind='stable')[:, :k]
    d = np.sqrt(np.maximum(2.0 - 2.0 * np.take_along_axis(np.minimum(cos, 1.0), nb, axis=1), 0.0))
    d[:, 0] = 0.0
    m = csr_matrix((d.reshape(-1).astype(np.float32), (np.repeat(np.arange(n), k), nb.reshape(-1))), shape=(n, n))
    ref = DBSCAN(n_jobs=1, min_samples=max(1, n // (max_vocab * 4)), metric='precomputed', eps=float(np.mean(d.astype(np.float32)))).fit_predict(m)
    n_good = len(set(ref.tolist()) - {-1})
    noise = ref < 0
    want = np.where(noise, n_good + np.cumsum(noise) - 1, ref)
    assert clusters == [int(c) for c in want]
    # planted structure: the members of a planted cluster share one id, different planted clusters do not
    got = np.asarray(clusters)
    for c in range(25):
        ids = set(got[owner == c].tolist())
        assert len(ids) == 1, (c, ids)
    assert len(set(got[owner < 25].tolist())) == 25
    # centroid of a planted cluster = normalised mean of its members
    c0 = got[owner == 0][0]
    mean = v[owner == 0].mean(axis=0)
    np.testing.assert_allclose(centers[c0], mean / np.linalg.norm(mean), atol=1e-5)


def test_output_onehot_size_shrinks_the_target_vocabulary():
    """prepare_vocabulary_and_word_vectors with max_vocabulary_size (reference :1035-1039): words of one cluster share a
    row of the (smaller) vocabulary matrix; flag rows and the end-of-sentence row keep their places behind the clusters."""
    from conv1d_text_vae_amd import Conv1dTextVAE
    from conv1d_text_vae_amd import embeddings as E
    rng = np.random.RandomState(5)
    v, owner = _planted_vectors(rng, n_clusters=12, per=10, n_lonely=8, dim=32)
    words = ['w%d' % i for i in range(v.shape[0])]
    kv = E.FastTextKeyedVectors(32, 0, 0, 0)
    kv.add(words, v)
    texts = tuple(tuple(words[i:i + 7]) for i in range(0, len(words), 7))
    full_vocab, full_mat = Conv1dTextVAE.prepare_vocabulary_and_word_vectors(texts, kv, None)
    vocab, mat = Conv1dTextVAE.prepare_vocabulary_and_word_vectors(texts, kv, None, 30, False)
    assert full_mat.shape[0] == len(words) + 2 and mat.shape[0] < full_mat.shape[0] and mat.shape[1] == 34
    assert vocab[''] == mat.shape[0] - 1 and mat[-1, -1] == 1.0 and mat[-2, 32] == 1.0
    np.testing.assert_allclose(np.linalg.norm(mat, axis=1), 1.0, atol=1e-5)
    for c in range(12):
        rows = set(vocab[words[i]] for i in np.nonzero(owner == c)[0])
        assert len(rows) == 1
