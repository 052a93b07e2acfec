"""Host-side fit-loop policy of the reference (conv1d_text_vae.py:308-337, :354-383).

The reference hands these decisions to three Keras 2.2 callbacks, run in this order at the end of every epoch:
`EarlyStopping(patience=min(5, max_epochs))`, `ReduceLROnPlateau('val_loss', factor=0.1, patience=2)` and
`ModelCheckpoint(save_best_only=True, save_weights_only=True)`.  Their update rules (Keras defaults: `min_delta` 0 for
early stopping, 1e-4 for the plateau rule, `cooldown` 0, `min_lr` 0, mode "min" for a loss) are restated here as plain
Python so that one loop drives both training stages; there is no numerical work in this file.
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional

import numpy as np


class EarlyStopping:
    def __init__(self, patience: int, min_delta: float = 0.0):
        self.patience, self.min_delta = int(patience), float(min_delta)
        self.best, self.wait, self.stop = math.inf, 0, False

    def on_epoch_end(self, val_loss: float):
        if val_loss - self.min_delta < self.best:
            self.best, self.wait = val_loss, 0
        else:
            self.wait += 1
            if self.wait >= self.patience:
                self.stop = True


class ReduceLROnPlateau:
    def __init__(self, get_lr: Callable[[], float], set_lr: Callable[[float], None], factor: float = 0.1,
                 patience: int = 2, min_delta: float = 1e-4, cooldown: int = 0, min_lr: float = 0.0):
        self.get_lr, self.set_lr = get_lr, set_lr
        self.factor, self.patience, self.min_delta = float(factor), int(patience), float(min_delta)
        self.cooldown, self.min_lr = int(cooldown), float(min_lr)
        self.best, self.wait, self.cooldown_counter = math.inf, 0, 0
        self.reductions: List[float] = []

    def on_epoch_end(self, val_loss: float):
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.wait = 0
        if val_loss < self.best - self.min_delta:
            self.best, self.wait = val_loss, 0
        elif self.cooldown_counter <= 0:
            self.wait += 1
            if self.wait >= self.patience:
                old_lr = float(self.get_lr())
                if old_lr > self.min_lr:
                    new_lr = max(old_lr * self.factor, self.min_lr)
                    self.set_lr(new_lr)
                    self.reductions.append(new_lr)
                    self.cooldown_counter = self.cooldown
                    self.wait = 0


class BestWeights:
    """ModelCheckpoint(save_best_only=True): keeps a copy of the weights of the best epoch (in memory instead of the
    reference's temporary HDF5 file, :312-316, :331-337)."""

    def __init__(self, snapshot: Callable[[], object]):
        self.snapshot = snapshot
        self.best, self.weights = math.inf, None

    def on_epoch_end(self, val_loss: float):
        if val_loss < self.best:
            self.best, self.weights = val_loss, self.snapshot()


def run_epochs(n_train_batches: int, n_eval_batches: int, max_epochs: int, train_batch: Callable[[int], None],
               eval_batch: Callable[[int], None], epoch_losses: Callable[[], tuple], get_lr, set_lr, snapshot, restore,
               verbose=False, rng: Optional[np.random.RandomState] = None, what: str = 'VAE'):
    """`fit_generator(shuffle=True, validation_data=..., callbacks=...)` (:323-330): per epoch the ORDER OF BATCH INDICES
    is shuffled (Keras shuffles a Sequence's index array, not the samples inside a batch), every batch trains once, the
    evaluation batches run in order, then the callbacks see the mean validation loss.  `train_batch(i)` / `eval_batch(i)`
    enqueue device work and accumulate losses on the device; `epoch_losses()` is the only host synchronisation of an
    epoch and returns (mean train loss, mean val loss).  Returns the per-epoch history."""
    if n_train_batches < 1:
        raise ValueError('There are no full batches for training: the training part of the data set is smaller than '
                         '`batch_size`.')
    if n_eval_batches < 1:
        raise ValueError('There are no full batches for evaluation: the validation part of the data set is smaller '
                         'than `batch_size`.')
    rng = rng if rng is not None else np.random
    stopper = EarlyStopping(patience=min(5, max_epochs))
    plateau = ReduceLROnPlateau(get_lr, set_lr, factor=0.1, patience=2)
    best = BestWeights(snapshot)
    history = []
    for epoch in range(max_epochs):
        order = np.arange(n_train_batches)
        rng.shuffle(order)
        for i in order:
            train_batch(int(i))
        for i in range(n_eval_batches):
            eval_batch(i)
        loss, val_loss = epoch_losses()
        history.append({'epoch': epoch + 1, 'loss': loss, 'val_loss': val_loss, 'lr': float(get_lr())})
        if verbose:
            print('{0} epoch {1}/{2}: loss = {3:.6f}, val_loss = {4:.6f}, lr = {5:g}'.format(
                what, epoch + 1, max_epochs, loss, val_loss, float(get_lr())))
        stopper.on_epoch_end(val_loss)
        plateau.on_epoch_end(val_loss)
        best.on_epoch_end(val_loss)
        if stopper.stop:
            if verbose:
                print('{0}: early stopping after epoch {1}.'.format(what, epoch + 1))
            break
    if best.weights is not None:
        restore(best.weights)
    return history


def run_epochs_dp(dp, global_batch: int, n_train_batches: int, n_eval_batches: int, max_epochs: int,
                  train_slice: Callable[[int, int], None], eval_slice: Callable[[int, int], None],
                  local_sums: Callable[[], np.ndarray], get_lr, set_lr, snapshot, restore, verbose=False,
                  seed: Optional[int] = None, what: str = 'VAE'):
    """`run_epochs` for `dp.world` data-parallel ranks (dist.DataParallel; world 1 = the plain loop).

    `global_batch` samples make one optimizer step, as in the reference (:323-330); rank r computes samples
    [first, first + count) = `dp.batch_slice(global_batch)` of every batch: `train_slice(first_sample, count)` /
    `eval_slice(...)` receive ABSOLUTE sample offsets into the training / evaluation set.  `local_sums()` returns this
    rank's (train loss sum, train batches, val loss sum, val batches) of the epoch and clears them; every loss term of a
    rank already carries 1/B_global, so the SUM over the ranks is the full-batch loss.  The sums are all-reduced once per
    epoch, hence every rank sees the same `val_loss` and takes the same EarlyStopping / ReduceLROnPlateau / best-weights
    decision (:308-316); the batch order of an epoch comes from one seed broadcast by rank 0; only rank 0 prints."""
    first, count = dp.batch_slice(global_batch)
    shared_seed = dp.broadcast_int(int(np.random.randint(0, 2 ** 31 - 1)) if seed is None else int(seed))
    rng = np.random.RandomState(shared_seed)

    def train_batch(i):
        train_slice(i * global_batch + first, count)

    def eval_batch(i):
        eval_slice(i * global_batch + first, count)

    def epoch_losses():
        a = dp.sum_over_ranks(local_sums())
        return float(a[0] / max(a[1] / dp.world, 1.0)), float(a[2] / max(a[3] / dp.world, 1.0))

    return run_epochs(n_train_batches, n_eval_batches, max_epochs, train_batch, eval_batch, epoch_losses, get_lr, set_lr,
                      snapshot, restore, verbose=bool(verbose) and dp.rank == 0, rng=rng, what=what)
