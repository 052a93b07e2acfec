"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend `nccl` is RCCL on ROCm, `gloo` in CPU tests).

The reference has no distributed code at all; these are the two places where the hot path shards:
  * training: contiguous split of the global batch, ONE all-reduce(SUM) of the flat fp32 gradient per step
    (every loss term already carries 1/B_global, so the sum IS the full-batch gradient);
  * nearest-word decode: contiguous split of the vocabulary rows, all-gather of the per-shard (score, index) pairs and a
    merge in which the highest score wins and ties go to the lowest global index (np.argmax / gensim argsort order).
"""
from __future__ import annotations

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
