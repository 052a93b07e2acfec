"""World-size-2 `gloo` tests (CPU) of the two sharded paths.  The local compute of each rank is
the NumPy oracle here -- what is under test is the sharding, the collectives and the merge rules that the GPU ranks use
unchanged (backend `nccl` = RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conv1d_text_vae_amd.dist import ShardedVocabDecoder, make_all_reduce, merge_top1, shard_range
from oracle import vae_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)


def test_shard_range_partitions():
    for n in (0, 1, 7, 8, 250001):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def test_merge_top1_ties_go_to_lowest_index():
    s = torch.tensor([[1.0, 0.5, 0.25], [1.0, 0.75, 0.25]])
    i = torch.tensor([[7, 1, 9], [3, 2, 4]], dtype=torch.int32)
    best, idx = merge_top1(s, i)
    assert best.tolist() == [1.0, 0.75, 0.25] and idx.tolist() == [3, 2, 4]


def _vocab_worker(rank, world, port, ret):
    _init(rank, world, port)
    rng = np.random.RandomState(0)
    Wn = rng.randn(101, 12).astype(np.float32)
    Wn /= np.linalg.norm(Wn, axis=1, keepdims=True)
    Wn[60] = Wn[10]                               # an exact tie across the two shards: index 10 must win
    q = np.concatenate([Wn[[10, 60, 99, 0]], rng.randn(13, 12).astype(np.float32)])
    b, e = shard_range(Wn.shape[0], world, rank)

    def local_fn(queries):
        s, i = O.vocab_argmax(queries.numpy(), Wn[b:e])
        return torch.from_numpy(s), torch.from_numpy(i + b)
    dec = ShardedVocabDecoder(Wn, q.shape[0], local_fn=local_fn)
    assert (dec.begin, dec.end) == (b, e)
    score, idx = dec.run(torch.from_numpy(q))
    ref_s, ref_i = O.vocab_argmax(q, Wn)
    ok = np.array_equal(idx.numpy(), ref_i) and np.allclose(score.numpy(), ref_s, atol=1e-6) and int(idx[1]) == 10
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_sharded_vocab_decode_matches_unsharded():
    world = 2
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_vocab_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        assert all(ret[r] for r in range(world))


def _dp_worker(rank, world, port, ret):
    """Data-parallel gradient: each rank differentiates its slice with every loss term scaled by 1/B_global (the oracle's
    local-mean gradient times B_local/B_global), the flat gradients are all-reduced (SUM) and must equal the full-batch
    gradient; the clip norm is then computed from the reduced gradient identically on every rank."""
    _init(rank, world, port)
    c = dict(B=6, T_in=12, T_out=11, E=8, n_filters=(6,), k=3, D=5, N=17)
    V = c['E'] + 2
    og = O.VAEGeometry(c['T_in'], c['T_out'], V, V, c['n_filters'], c['k'], c['D'])
    params = O.init_params(og, seed=3, dtype=np.float64)
    Wv = O.synthetic_vocab(c['N'] - 2, c['E'], seed=5, dtype=np.float64)
    x, ids = O.synthetic_batch(Wv, Wv.shape[0], c['B'], c['T_in'], c['T_out'], seed=7)
    # RNG keyed by the GLOBAL element index: a rank draws the rows of its slice from the same stream
    D, flat = c['D'], og.flat_size
    eps = O.philox_normal(c['B'] * D, seed=11, stream=1).reshape(c['B'], D).astype(np.float64)
    me = O.philox_bernoulli(c['B'] * flat, 0.7, seed=11, stream=2).reshape(c['B'], flat).astype(bool)
    md = O.philox_bernoulli(c['B'] * D, 0.7, seed=11, stream=3).reshape(c['B'], D).astype(bool)
    b, e = shard_range(c['B'], world, rank)
    Bl = e - b
    eps_l = O.philox_normal(Bl * D, seed=11, stream=1, offset=b * D).reshape(Bl, D).astype(np.float64)
    me_l = O.philox_bernoulli(Bl * flat, 0.7, seed=11, stream=2, offset=b * flat).reshape(Bl, flat).astype(bool)
    ok = np.array_equal(eps_l, eps[b:e]) and np.array_equal(me_l, me[b:e])
    _, g_loc, _ = O.train_forward_backward(og, params, Wv, x[b:e], ids[b:e], eps[b:e], me[b:e], md[b:e])
    names = sorted(g_loc)
    flat_g = torch.from_numpy(np.concatenate([g_loc[n].reshape(-1) for n in names]) * (Bl / c['B']))
    make_all_reduce()(flat_g)
    _, g_full, _ = O.train_forward_backward(og, params, Wv, x, ids, eps, me, md)
    ref = np.concatenate([g_full[n].reshape(-1) for n in names])
    ok = ok and np.allclose(flat_g.numpy(), ref, rtol=1e-9, atol=1e-12)
    norm = float(torch.sqrt((flat_g * flat_g).sum()))
    ok = ok and abs(norm - O.global_norm(g_full)) < 1e-9
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_data_parallel_gradient_equals_full_batch_gradient():
    world = 2
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_dp_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        assert all(ret[r] for r in range(world))


def test_bench_spawns_its_own_ranks_and_relays_one_json_line():
    """`python bench.py --gpus 2` without torchrun (how the round-end driver starts it): the parent must start both ranks
    before touching a GPU, the ranks must find each other on 127.0.0.1, and stdout must carry exactly rank 0's line.
    `--dry-run` swaps the GPU work for a gloo all-reduce, so this runs without a GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--dry-run'], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec == {'metric': 'dry_run', 'n_gpus': 2, 'rank_sum': 3.0}


# ---- data-parallel fit loop (training.run_epochs_dp + dist.DataParallel) ---------------------------------------------
def _fit_case():
    c = dict(B=4, T_in=10, T_out=9, E=6, n_filters=(5,), k=3, D=4, N=15, n_train=12, n_eval=4)
    V = c['E'] + 2
    og = O.VAEGeometry(c['T_in'], c['T_out'], V, V, c['n_filters'], c['k'], c['D'])
    Wv = O.synthetic_vocab(c['N'] - 2, c['E'], seed=5, dtype=np.float64)
    x, ids = O.synthetic_batch(Wv, Wv.shape[0], c['n_train'] + c['n_eval'], c['T_in'], c['T_out'], seed=7)
    return c, og, Wv, x.astype(np.float64), ids


def _run_fit(dp, max_epochs=20, lr0=0.3):
    """The estimator's VAE stage with the NumPy oracle as the local compute of a rank: per step the rank differentiates
    its slice of the global batch (loss terms scaled by 1/B_global), the flat gradient is all-reduced, Nadam runs
    replicated.  The large learning rate makes the validation loss go up within a few epochs, so the plateau rule and
    the early stop both fire inside `max_epochs`."""
    from conv1d_text_vae_amd.training import run_epochs_dp
    c, og, Wv, x, ids = _fit_case()
    B, D, flat = c['B'], c['D'], og.flat_size
    params = O.init_params(og, seed=3, dtype=np.float64)
    st = O.NadamState()
    lr = [lr0]
    acc = np.zeros(4)
    names = sorted(params)
    step = [0]
    xt, it, xe, ie = x[:c['n_train']], ids[:c['n_train']], x[c['n_train']:], ids[c['n_train']:]

    def draws(first, count, stream0):
        # keyed by (step, GLOBAL sample index): any split of the batch over ranks draws the same numbers
        off = step[0] * (c['n_train'] + c['n_eval'])
        eps = O.philox_normal(count * D, seed=11, stream=stream0, offset=(off + first) * D).reshape(count, D).astype(np.float64)
        me = O.philox_bernoulli(count * flat, 0.7, seed=11, stream=stream0 + 1, offset=(off + first) * flat).reshape(count, flat).astype(bool)
        md = O.philox_bernoulli(count * D, 0.7, seed=11, stream=stream0 + 2, offset=(off + first) * D).reshape(count, D).astype(bool)
        return eps, me, md

    def train_slice(first, count):
        eps, me, md = draws(first, count, 1)
        sl = slice(first, first + count)
        loss_l, g_loc, _ = O.train_forward_backward(og, params, Wv, xt[sl], it[sl], eps, me, md)
        flat_g = torch.from_numpy(np.concatenate([g_loc[n].reshape(-1) for n in names]) * (count / B))
        if dp.world > 1:
            dp.all_reduce_async(flat_g).wait()
        g, o = {}, 0
        for n in names:
            g[n] = flat_g.numpy()[o:o + params[n].size].reshape(params[n].shape)
            o += params[n].size
        O.nadam_step(params, g, st, lr=lr[0])
        acc[0] += loss_l * count / B
        acc[1] += 1
        step[0] += 1

    def eval_slice(first, count):
        eps, _, _ = draws(first, count, 4)
        sl = slice(first, first + count)
        acc[2] += O.eval_loss(og, params, Wv, xe[sl], ie[sl], eps) * count / B
        acc[3] += 1

    def local_sums():
        a = acc.copy()
        acc[:] = 0
        return a

    snap = lambda: {k: v.copy() for k, v in params.items()}

    def restore(w):
        for k in params:
            params[k][...] = w[k]
    hist = run_epochs_dp(dp, B, c['n_train'] // B, c['n_eval'] // B, max_epochs, train_slice, eval_slice, local_sums,
                         lambda: lr[0], lambda v: lr.__setitem__(0, float(v)), snap, restore, seed=1234)
    return hist, np.concatenate([params[n].reshape(-1) for n in names])


def _fit_worker(rank, world, port, ret):
    from conv1d_text_vae_amd.dist import DataParallel
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dp = DataParallel.from_env(use_gpu=False)
    assert (dp.world, dp.rank) == (world, rank) and dp.batch_slice(4) == (2 * rank, 2)
    assert dp.broadcast_int(100 + rank) == 100
    hist, w = _run_fit(dp)
    ret[rank] = (hist, w)
    dist.destroy_process_group()


def test_data_parallel_fit_loop_takes_the_same_decisions_on_every_rank():
    """World-size-2 `gloo`: training.run_epochs_dp drives the fit loop of the reference (:323-330 with the callbacks of
    :308-316) over two ranks that each compute half of every batch.  Both ranks must see the same epoch losses, hence the
    same stop epoch, the same learning-rate history and the same restored best weights -- and all of it must equal the
    single-process loop over whole batches."""
    from conv1d_text_vae_amd.dist import DataParallel
    world = 2
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_fit_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        (h0, w0), (h1, w1) = ret[0], ret[1]
    assert h0 == h1 and np.array_equal(w0, w1)
    href, wref = _run_fit(DataParallel())
    assert [e['epoch'] for e in h0] == [e['epoch'] for e in href]             # same stop epoch
    assert [e['lr'] for e in h0] == [e['lr'] for e in href]                   # same plateau decisions
    assert len(set(e["lr"] for e in href)) > 1 and len(href) < 20, 'the case must exercise the plateau rule and the early stop'
    for a, b in zip(h0, href):
        assert abs(a['loss'] - b['loss']) < 1e-9 and abs(a['val_loss'] - b['val_loss']) < 1e-9
    np.testing.assert_allclose(w0, wref, rtol=1e-9, atol=1e-12)


def _topn_worker(rank, world, port, ret):
    from conv1d_text_vae_amd.dist import ShardedVocabTopN
    _init(rank, world, port)
    rng = np.random.RandomState(0)
    Wn = rng.randn(57, 10).astype(np.float32)
    Wn /= np.linalg.norm(Wn, axis=1, keepdims=True)
    Wn[40] = Wn[3]                                  # a tie across the shards: the lower index comes first
    q = np.concatenate([Wn[[3, 40, 56]], rng.randn(6, 10).astype(np.float32)])
    b, e = shard_range(Wn.shape[0], world, rank)

    def local_fn(queries, topn):
        s = queries @ Wn[b:e].T
        order = np.lexsort((np.broadcast_to(np.arange(e - b), s.shape), -s), axis=1)[:, :topn]
        return np.take_along_axis(s, order, axis=1), (order + b).astype(np.int32)
    top = ShardedVocabTopN(Wn, local_fn=local_fn)
    score, idx = top.run(q, 5)
    s = q @ Wn.T
    ref = np.lexsort((np.broadcast_to(np.arange(Wn.shape[0]), s.shape), -s), axis=1)[:, :5]
    ret[rank] = bool(np.array_equal(idx, ref) and np.allclose(score, np.take_along_axis(s, ref, axis=1), atol=1e-6)
                     and idx[0, 0] == 3 and idx[0, 1] == 40)
    dist.destroy_process_group()


def test_sharded_topn_matches_unsharded():
    world = 2
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_topn_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        assert all(ret[r] for r in range(world))


def _kv_worker(rank, world, port, ret):
    from conv1d_text_vae_amd.dist import ShardedVocabTopN
    from conv1d_text_vae_amd.embeddings import FastTextKeyedVectors
    _init(rank, world, port)
    rng = np.random.RandomState(1)
    kv = FastTextKeyedVectors(8, 3, 5)
    words = ['w%02d' % i for i in range(33)]
    kv.add(words, rng.randn(33, 8).astype(np.float32))
    Wn = kv.vectors_norm
    b, e = shard_range(Wn.shape[0], world, rank)

    def local_fn(queries, topn):            # (the rank's rows ranked on the CPU: the HIP VocabTopN needs a GPU)
        s = queries @ Wn[b:e].T
        order = np.lexsort((np.broadcast_to(np.arange(e - b), s.shape), -s), axis=1)[:, :topn]
        return np.take_along_axis(s, order, axis=1), (order + b).astype(np.int32)
    kv.shard_over_ranks = True
    kv._topn = ShardedVocabTopN(Wn, device='cpu', local_fn=local_fn)
    q = rng.randn(5, 8).astype(np.float32)
    got = kv.similar_by_vectors(q, topn=4, device='cpu')
    s = (q / np.linalg.norm(q, axis=1, keepdims=True)) @ Wn.T
    ref = np.argsort(-s, axis=1, kind='stable')[:, :4]
    ret[rank] = [[w for w, _ in row] for row in got] == [[words[j] for j in row] for row in ref]
    dist.destroy_process_group()


def test_keyed_vectors_similar_by_vectors_over_a_sharded_table():
    """FastTextKeyedVectors.shard_over_ranks: `similar_by_vectors` (find_best_words' search, reference :584-587) as a
    collective over two ranks, each ranking half of the table, returns the words of the unsharded ranking on every rank."""
    world = 2
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_kv_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        assert all(ret[r] for r in range(world))
