"""Running metrics with the two-method protocol of ``allennlp.training.metrics`` that the
reference's models use (reference: probnmn/models/nmn.py:121-124,262-263,292-294):
``metric(values...)`` to update, ``metric.get_metric(reset)`` to read."""
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
