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
