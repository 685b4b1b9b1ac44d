"""Host-side running metrics (probnmn.running_metrics)."""
import math

import torch

from probnmn.running_metrics import BLEU, Average, BooleanAccuracy


def test_bleu_known_answer():
    """Hand computation of allennlp 0.9.0's corpus BLEU (clipped n-gram counts summed over the batch,
    n-grams containing an excluded index dropped, brevity penalty on non-excluded lengths)."""
    b = BLEU(exclude_indices={0, 2, 3})
    pred = torch.tensor([[5, 6, 7, 8, 3, 0], [5, 5, 9, 3, 0, 0]])
    gold = torch.tensor([[2, 5, 6, 7, 9, 3], [2, 5, 9, 9, 3, 0]])
    b(pred, gold)
    # unigrams 5/7, bigrams 3/5, trigrams 1/3, 4-grams 0/1; lengths 7 vs 7 -> brevity penalty 1
    want = math.exp(0.25 * sum(math.log(m + 1e-13) - math.log(t + 1e-13) for m, t in ((5, 7), (3, 5), (1, 3), (0, 1))))
    assert b.get_metric(reset=True)["BLEU"] == want
    b(gold, gold)
    assert abs(b.get_metric()["BLEU"] - 1.0) < 1e-9
    short = BLEU(exclude_indices={0})
    short(torch.tensor([[5, 6, 7, 8, 0, 0, 0, 0]]), torch.tensor([[5, 6, 7, 8, 9, 9, 9, 9]]))
    assert abs(short.get_metric()["BLEU"] - math.exp(1 - 8 / 4)) < 1e-9  # perfect precision, half the length


def test_average_and_boolean_accuracy():
    a = Average()
    a(2.0), a(torch.tensor(4.0))
    assert a.get_metric(reset=True) == 3.0 and a.get_metric() == 0.0
    acc = BooleanAccuracy()
    acc(torch.tensor([1, 2, 3, 28]), torch.tensor([1, 0, 3, 5]))
    assert acc.get_metric() == 0.5
