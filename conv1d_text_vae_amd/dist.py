"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend `nccl` is RCCL on ROCm, `gloo` in CPU tests).

The reference has no distributed code at all; these are the two places where the hot path shards:
  * training: contiguous split of the global batch, ONE all-reduce(SUM) of the flat fp32 gradient per step
    (every loss term already carries 1/B_global, so the sum IS the full-batch gradient);
  * nearest-word decode: contiguous split of the vocabulary rows, all-gather of the per-shard (score, index) pairs and a
    merge in which the highest score wins and ties go to the lowest global index (np.argmax / gensim argsort order).
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Tuple

import numpy as np
import torch


def shard_range(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) of `n` items for `rank` (the first n % world ranks get one more)."""
    q, r = divmod(int(n), int(world))
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


def merge_top1(scores: torch.Tensor, idxs: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """scores, idxs: [R, M].  Highest score wins, ties -> lowest index.  Device-agnostic (plumbing on tiny tensors;
    on the GPU the same rule is also available as c1v_argmax_merge)."""
    best = scores.max(dim=0).values
    cand = torch.where(scores == best.unsqueeze(0), idxs, torch.full_like(idxs, torch.iinfo(idxs.dtype).max))
    return best, cand.min(dim=0).values


def make_all_reduce(group=None) -> Callable[[torch.Tensor], None]:
    import torch.distributed as dist

    def all_reduce(t: torch.Tensor) -> None:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return all_reduce


class ShardedVocabDecoder:
    """Nearest vocabulary row with the vocabulary rows split over the ranks of `group`.

    `local_fn(queries) -> (score [M] f32, global index [M] i32)` computes the arg-max over THIS rank's rows; by default
    it is the HIP `VocabDecoder` (fused MFMA GEMM + arg-max) built over rows [begin, end) with n_base = begin."""

    def __init__(self, vocab_norm: np.ndarray, max_queries: int, dtype: str = 'bf16', device: Optional[str] = None,
                 group=None, local_fn=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.begin, self.end = shard_range(vocab_norm.shape[0], self.world, self.rank)
        self.device = device
        if local_fn is None:
            from .engine import VocabDecoder
            dec = VocabDecoder(vocab_norm[self.begin:self.end], max_queries, dtype=dtype, device=device or 'cuda:0',
                               n_base=self.begin)

            def local_fn(q):
                dec.load_queries(q)
                return dec.run()
        self.local_fn = local_fn

    def run(self, queries: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        score, idx = self.local_fn(queries)
        if self.world == 1:
            return score, idx
        score = score.contiguous()
        idx = idx.contiguous().to(torch.int32)
        all_s = [torch.empty_like(score) for _ in range(self.world)]
        all_i = [torch.empty_like(idx) for _ in range(self.world)]
        self.dist.all_gather(all_s, score, group=self.group)       # 8 bytes per query per rank
        self.dist.all_gather(all_i, idx, group=self.group)
        return merge_top1(torch.stack(all_s), torch.stack(all_i))


def merge_topn(scores: np.ndarray, idxs: np.ndarray, topn: int) -> Tuple[np.ndarray, np.ndarray]:
    """scores, idxs: [R, M, n] per-shard top-n lists (best first).  The global top-n of every query: highest score first,
    ties -> lowest global index (gensim's argsort order over the unsharded table)."""
    R, M, n = scores.shape
    s = np.transpose(scores, (1, 0, 2)).reshape(M, R * n)
    i = np.transpose(idxs, (1, 0, 2)).reshape(M, R * n)
    order = np.lexsort((i, -s), axis=1)[:, :topn]
    return np.take_along_axis(s, order, axis=1), np.take_along_axis(i, order, axis=1)


class ShardedVocabTopN:
    """`similar_by_vector(v, topn)` (reference :587) with the table rows split over the ranks: every rank ranks its own
    rows (`local_fn(queries, topn) -> (score [M, n], GLOBAL index [M, n])`, by default the HIP `VocabTopN` over rows
    [begin, end) with n_base = begin), the per-shard lists are all-gathered (8 bytes x topn per query and rank) and
    merged with `merge_topn`.  Collective: every rank must call `run` with the same queries."""

    def __init__(self, vocab_norm: np.ndarray, device: Optional[str] = None, group=None, local_fn=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.begin, self.end = shard_range(vocab_norm.shape[0], self.world, self.rank)
        self.n_total = int(vocab_norm.shape[0])
        self.device_str = device
        if local_fn is None:
            from .engine import VocabTopN
            top = VocabTopN(vocab_norm[self.begin:self.end], device=device or 'cuda:0')
            begin = self.begin

            def local_fn(q, topn):
                return top.run(q, topn, n_base=begin)
        self.local_fn = local_fn

    def run(self, queries: np.ndarray, topn: int, n_base: int = 0):
        n_loc = int(min(topn, self.end - self.begin))
        q = np.asarray(queries, dtype=np.float32)
        M = q.shape[0]
        score = np.full((M, topn), -np.inf, dtype=np.float32)
        idx = np.full((M, topn), np.iinfo(np.int32).max, dtype=np.int32)
        if n_loc > 0:
            s, i = self.local_fn(q, n_loc)
            score[:, :n_loc], idx[:, :n_loc] = s, i
        if self.world > 1:
            backend = self.dist.get_backend(self.group)
            dev = torch.device(self.device_str or 'cuda:0') if backend == 'nccl' else torch.device('cpu')
            ts, ti = torch.from_numpy(score).to(dev), torch.from_numpy(idx).to(dev)
            all_s = [torch.empty_like(ts) for _ in range(self.world)]
            all_i = [torch.empty_like(ti) for _ in range(self.world)]
            self.dist.all_gather(all_s, ts, group=self.group)
            self.dist.all_gather(all_i, ti, group=self.group)
            score, idx = merge_topn(torch.stack(all_s).cpu().numpy(), torch.stack(all_i).cpu().numpy(), topn)
        return score, idx + np.int32(n_base)


class DataParallel:
    """The ranks of a data-parallel `fit()`: one process per GPU, started by `torchrun` (WORLD_SIZE / RANK / LOCAL_RANK /
    MASTER_ADDR / MASTER_PORT in the environment), backend `nccl` = RCCL over xGMI (`gloo` for the CPU tests).
    World size 1 (no torchrun) is the plain single-process fit and touches torch.distributed nowhere.

    The reference has no counterpart (it is single-process).  Semantics kept from it: `batch_size` stays the number of
    samples of ONE optimizer step -- the GLOBAL batch; rank r computes samples [r*B/world, (r+1)*B/world) of every batch."""

    def __init__(self, world: int = 1, rank: int = 0, local_rank: int = 0, group=None, device: Optional[torch.device] = None):
        self.world, self.rank, self.local_rank, self.group = int(world), int(rank), int(local_rank), group
        self.dev = device if device is not None else torch.device('cpu')      # where the collectives' tensors live

    @classmethod
    def from_env(cls, use_gpu: bool = True) -> 'DataParallel':
        world = int(os.environ.get('WORLD_SIZE', '1'))
        if world <= 1:
            return cls()
        import torch.distributed as dist
        local_rank = int(os.environ.get('LOCAL_RANK', os.environ.get('RANK', '0')))
        if use_gpu:
            torch.cuda.set_device(local_rank)
            dev = torch.device('cuda', local_rank)
        else:
            dev = torch.device('cpu')
        if not dist.is_initialized():
            if use_gpu:
                os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
                dist.init_process_group('nccl', device_id=dev)
            else:
                dist.init_process_group('gloo')
        return cls(dist.get_world_size(), dist.get_rank(), local_rank, None, dev)

    @property
    def device_str(self) -> str:
        return 'cuda:%d' % self.local_rank

    def batch_slice(self, global_batch: int) -> Tuple[int, int]:
        """(first sample, samples) of this rank inside a global batch."""
        if global_batch % self.world != 0:
            raise ValueError('`batch_size` ({0}) must be a multiple of the number of data-parallel ranks ({1}).'.format(
                global_batch, self.world))
        bl = global_batch // self.world
        return self.rank * bl, bl

    def all_reduce_async(self, t: torch.Tensor):
        """SUM over the ranks, asynchronous (the engine overlaps it with the rest of backward)."""
        import torch.distributed as dist
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def sum_over_ranks(self, values) -> np.ndarray:
        """float64 vector summed over the ranks (epoch loss sums: tiny, once per epoch)."""
        a = np.asarray(values, dtype=np.float64)
        if self.world == 1:
            return a
        import torch.distributed as dist
        t = torch.from_numpy(a.copy()).to(self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def broadcast_int(self, value: int) -> int:
        """Rank 0's value on every rank (seeds: weights, shuffles and dropout streams must agree)."""
        if self.world == 1:
            return int(value)
        import torch.distributed as dist
        t = torch.tensor([int(value)], dtype=torch.int64, device=self.dev)
        dist.broadcast(t, src=0, group=self.group)
        return int(t.item())

    def all_gather_rows(self, t: torch.Tensor) -> torch.Tensor:
        """[n, ...] per rank -> [world*n, ...] in rank order (the stage-2 generator trains replicated on whole batches)."""
        if self.world == 1:
            return t
        import torch.distributed as dist
        t = t.contiguous()
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t, group=self.group)
        return out
