"""Running metrics with the two-method protocol of ``allennlp.training.metrics`` that the
reference's models use (reference: probnmn/models/nmn.py:121-124,262-263,292-294):
``metric(values...)`` to update, ``metric.get_metric(reset)`` to read."""
import math
from collections import Counter
from typing import Iterable, Optional, Sequence

import numpy as np
import torch


class Average:
    """Tensor updates are summed on their device and read back only in ``get_metric`` -- an update
    costs no host synchronisation (the reference's ``.item()`` per update does)."""

    def __init__(self):
        self._total = 0.0
        self._count = 0

    def __call__(self, value) -> None:
        if isinstance(value, torch.Tensor):
            self._total = self._total + value.detach().float().sum()
        else:
            self._total += float(value)
        self._count += 1

    def get_metric(self, reset: bool = False) -> float:
        value = float(self._total) / self._count if self._count else 0.0
        if reset:
            self.reset()
        return value

    def reset(self) -> None:
        self._total, self._count = 0.0, 0


class BooleanAccuracy:
    """Fraction of examples whose prediction equals the gold label (all trailing dims equal)."""

    def __init__(self):
        self._correct = 0.0
        self._total = 0.0

    def __call__(self, predictions: torch.Tensor, gold: torch.Tensor, mask=None) -> None:
        predictions, gold = predictions.detach(), gold.detach()
        eq = predictions.reshape(predictions.size(0), -1).eq(gold.reshape(gold.size(0), -1)).all(dim=1)
        if mask is not None:
            keep = mask.reshape(mask.size(0), -1).any(dim=1)
            eq = eq[keep]
        self._correct = self._correct + eq.sum()  # stays on the device until get_metric
        self._total += float(eq.numel())

    def get_metric(self, reset: bool = False) -> float:
        value = float(self._correct) / self._total if self._total else 0.0
        if reset:
            self.reset()
        return value

    def reset(self) -> None:
        self._correct, self._total = 0.0, 0.0


class BLEU:
    """Corpus BLEU as ``allennlp.training.metrics.BLEU`` (0.9.0) computes it -- what
    ``SimpleSeq2Seq(use_bleu=True)`` records in evaluation (reference: probnmn/modules/seq2seq_base.py:91,
    260,367): clipped n-gram matches and totals for n = 1..4 summed over every (prediction, target) pair,
    n-grams that contain an excluded index (padding, @start@, @end@) dropped, lengths counted over
    non-excluded tokens, brevity penalty ``exp(1 - reference_length / prediction_length)`` when the
    predictions are shorter, and ``exp(sum_n w_n (log(matches_n + 1e-13) - log(totals_n + 1e-13)))``.
    Evaluation-only and string-like work: it runs on the host over the (small) token matrices."""

    def __init__(self, ngram_weights: Iterable[float] = (0.25, 0.25, 0.25, 0.25),
                 exclude_indices: Optional[Sequence[int]] = None):
        self._ngram_weights = tuple(ngram_weights)
        self._exclude = set(exclude_indices or ())
        self._exclude_array = np.asarray(sorted(self._exclude), dtype=np.int64)
        self._base = 1 << 12  # keys: row, then n <= 4 tokens of 12 bits each (vocabularies here: < 100 entries)
        if len(self._ngram_weights) > 4:
            raise ValueError("n-grams beyond 4 tokens do not fit the 64-bit keys")
        self.reset()

    def reset(self) -> None:
        self._matches = Counter()
        self._totals = Counter()
        self._prediction_length = 0
        self._reference_length = 0
        self._pending = []  # device batches on their way to the host: (pinned predictions, pinned targets, event)

    # Device tensors are counted LATER: the matrices go to page-locked buffers with asynchronous copies, and a batch is
    # counted once its copy has landed -- when the next batch arrives, or in get_metric().  Counting at once made every
    # evaluation forward pass wait for its own kernels (a validation loop could queue nothing behind the generator's pass
    # until it had finished: evaluators.answering_evaluator).  The counts are the same sums in the same order.
    _RING = 4

    def _drain(self, wait: bool) -> None:
        while self._pending and (wait or self._pending[0][2].query()):
            pred, gold, event = self._pending.pop(0)
            event.synchronize()
            self._count(pred.numpy().astype(np.int64, copy=False), gold.numpy().astype(np.int64, copy=False))

    def _defer(self, predictions: torch.Tensor, gold_targets: torch.Tensor) -> None:
        if len(self._pending) >= self._RING:  # (its buffers are the oldest entry's: count that one first)
            pred, gold, event = self._pending.pop(0)
            event.synchronize()
            self._count(pred.numpy().astype(np.int64, copy=False), gold.numpy().astype(np.int64, copy=False))
        pool = self.__dict__.setdefault("_pinned", {})
        slot = self.__dict__["_slot"] = (self.__dict__.get("_slot", -1) + 1) % self._RING
        hosts = []
        for name, t in (("p", predictions), ("g", gold_targets)):
            key = (name, slot, tuple(t.shape), t.dtype)
            if key not in pool:
                pool[key] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            pool[key].copy_(t.detach(), non_blocking=True)
            hosts.append(pool[key])
        event = torch.cuda.Event()
        event.record()
        self._pending.append((hosts[0], hosts[1], event))

    def _keys(self, tokens: np.ndarray, n: int) -> np.ndarray:
        """One int64 key per n-gram of every row that holds no excluded index: (row, the n tokens)."""
        B, T = tokens.shape
        if T < n:
            return np.empty(0, np.int64)
        ok = np.ones((B, T - n + 1), bool)
        h = np.zeros((B, T - n + 1), np.int64)
        for k in range(n):
            part = tokens[:, k:T - n + 1 + k]
            if self._exclude:
                ok &= ~np.isin(part, self._exclude_array)
            h = h * self._base + part
        h += np.arange(B, dtype=np.int64)[:, None] * (self._base ** n)
        return h[ok]

    def __call__(self, predictions: torch.Tensor, gold_targets: torch.Tensor) -> None:
        # (vectorised over the batch: the per-row Python loops of the first version took 35 ms per 256-row validation
        # batch -- nine tenths of evaluate_answer_accuracy)
        if predictions.is_cuda and gold_targets.is_cuda and predictions.dim() == 2 and gold_targets.dim() == 2 \
                and predictions.size(0) == gold_targets.size(0):
            self._drain(wait=False)
            self._defer(predictions, gold_targets)
            return
        self._drain(wait=True)  # (counts stay in call order)
        self._count(predictions.detach().cpu().numpy().astype(np.int64, copy=False),
                    gold_targets.detach().cpu().numpy().astype(np.int64, copy=False))

    def _count(self, pred: np.ndarray, gold: np.ndarray) -> None:
        if pred.ndim != 2 or gold.ndim != 2 or pred.shape[0] != gold.shape[0]:
            raise ValueError("BLEU takes (batch, length) prediction and target matrices")
        top = int(max(pred.max(initial=0), gold.max(initial=0))) + 1
        if top > self._base:
            raise ValueError("token index %d beyond the metric's key base %d" % (top - 1, self._base))
        low = int(min(pred.min(initial=0), gold.min(initial=0)))
        if low < 0:  # (a negative id -- a -1 padding value, say -- would alias other n-grams' keys)
            raise ValueError("negative token index %d: the n-gram keys pack non-negative indices" % low)
        if pred.shape[0] * self._base ** len(self._ngram_weights) >= 1 << 63:
            # (row * base**n overflows int64 from 32 768 rows on: count such a batch in pieces)
            step = (1 << 62) // self._base ** len(self._ngram_weights)
            for lo in range(0, pred.shape[0], step):
                self._count(pred[lo:lo + step], gold[lo:lo + step])
            return
        for n in range(1, len(self._ngram_weights) + 1):
            pk, pc = np.unique(self._keys(pred, n), return_counts=True)
            gk, gc = np.unique(self._keys(gold, n), return_counts=True)
            _, pi, gi = np.intersect1d(pk, gk, assume_unique=True, return_indices=True)
            self._matches[n] += int(np.minimum(pc[pi], gc[gi]).sum())
            self._totals[n] += int(pc.sum())
        if not self._exclude:
            self._prediction_length += pred.size
            self._reference_length += gold.size
        else:
            self._prediction_length += int((~np.isin(pred, self._exclude_array)).sum())
            self._reference_length += int((~np.isin(gold, self._exclude_array)).sum())

    def _brevity_penalty(self) -> float:
        if self._prediction_length > self._reference_length:
            return 1.0
        if self._reference_length == 0 or self._prediction_length == 0:
            return 0.0
        return math.exp(1.0 - self._reference_length / self._prediction_length)

    def get_metric(self, reset: bool = False):
        self._drain(wait=True)
        scores = (w * (math.log(self._matches[n] + 1e-13) - math.log(self._totals[n] + 1e-13))
                  for n, w in enumerate(self._ngram_weights, start=1))
        bleu = self._brevity_penalty() * math.exp(sum(scores))
        if reset:
            self.reset()
        return {"BLEU": bleu}
