"""Running metrics with the two-method protocol of ``allennlp.training.metrics`` that the
reference's models use (reference: probnmn/models/nmn.py:121-124,262-263,292-294):
``metric(values...)`` to update, ``metric.get_metric(reset)`` to read."""
import math
from collections import Counter
from typing import Iterable, Optional, Sequence

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
        self.reset()

    def reset(self) -> None:
        self._matches = Counter()
        self._totals = Counter()
        self._prediction_length = 0
        self._reference_length = 0

    def _ngrams(self, row, n: int) -> Counter:
        out = Counter()
        for i in range(len(row) - n + 1):
            gram = tuple(row[i:i + n])
            if self._exclude and any(t in self._exclude for t in gram):
                continue
            out[gram] += 1
        return out

    def __call__(self, predictions: torch.Tensor, gold_targets: torch.Tensor) -> None:
        pred, gold = predictions.detach().cpu().tolist(), gold_targets.detach().cpu().tolist()
        for n in range(1, len(self._ngram_weights) + 1):
            for p_row, g_row in zip(pred, gold):
                p_counts, g_counts = self._ngrams(p_row, n), self._ngrams(g_row, n)
                self._matches[n] += sum(min(c, g_counts.get(gram, 0)) for gram, c in p_counts.items())
                self._totals[n] += sum(p_counts.values())
        if not self._exclude:
            self._prediction_length += sum(len(r) for r in pred)
            self._reference_length += sum(len(r) for r in gold)
        else:
            self._prediction_length += sum(1 for r in pred for t in r if t not in self._exclude)
            self._reference_length += sum(1 for r in gold for t in r if t not in self._exclude)

    def _brevity_penalty(self) -> float:
        if self._prediction_length > self._reference_length:
            return 1.0
        if self._reference_length == 0 or self._prediction_length == 0:
            return 0.0
        return math.exp(1.0 - self._reference_length / self._prediction_length)

    def get_metric(self, reset: bool = False):
        scores = (w * (math.log(self._matches[n] + 1e-13) - math.log(self._totals[n] + 1e-13))
                  for n, w in enumerate(self._ngram_weights, start=1))
        bleu = self._brevity_penalty() * math.exp(sum(scores))
        if reset:
            self.reset()
        return {"BLEU": bleu}
