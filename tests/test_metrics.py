"""Host-side running metrics (probnmn.running_metrics)."""
import math

import pytest
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


def test_bleu_vectorised_over_the_batch_equals_the_per_row_count():
    """The metric counts n-grams over whole (batch, length) matrices at once (numpy keys); this restates the count one
    row and one n-gram at a time, as allennlp 0.9.0's BLEU walks them, on ragged content with excluded indices."""
    import math
    from collections import Counter

    def per_row(pred, gold, exclude):
        m, t = Counter(), Counter()

        def grams(row, n):
            out = Counter()
            for i in range(len(row) - n + 1):
                g = tuple(row[i:i + n])
                if not any(x in exclude for x in g):
                    out[g] += 1
            return out
        for n in range(1, 5):
            for p, g in zip(pred, gold):
                pc, gc = grams(p, n), grams(g, n)
                m[n] += sum(min(c, gc.get(k, 0)) for k, c in pc.items())
                t[n] += sum(pc.values())
        pl = sum(1 for r in pred for x in r if x not in exclude)
        rl = sum(1 for r in gold for x in r if x not in exclude)
        bp = 1.0 if pl > rl else (0.0 if rl == 0 or pl == 0 else math.exp(1 - rl / pl))
        return bp * math.exp(sum(0.25 * (math.log(m[n] + 1e-13) - math.log(t[n] + 1e-13)) for n in range(1, 5)))

    g = torch.Generator().manual_seed(0)
    for exclude in ({0, 2, 3}, set()):
        pred = torch.randint(0, 12, (37, 26), generator=g)
        gold = torch.randint(0, 12, (37, 27), generator=g)
        pred[:, 20:] = 0
        b = BLEU(exclude_indices=exclude)
        b(pred[:20], gold[:20])  # (accumulates over calls)
        b(pred[20:], gold[20:])
        assert b.get_metric()["BLEU"] == pytest.approx(per_row(pred.tolist(), gold.tolist(), exclude), abs=1e-12)


def test_bleu_key_limits():
    """The packed int64 keys: a negative index is refused (it would alias another n-gram), and a batch whose
    row * base**4 would overflow is counted in pieces with the same result as row-sized calls."""
    b = BLEU(exclude_indices={0})
    with pytest.raises(ValueError):
        b(torch.tensor([[1, -1, 2]]), torch.tensor([[1, 2, 2]]))
    g = torch.Generator().manual_seed(1)
    pred = torch.randint(0, 6, (40000, 5), generator=g)
    gold = torch.randint(0, 6, (40000, 5), generator=g)
    whole, parts = BLEU(exclude_indices={0}), BLEU(exclude_indices={0})
    whole(pred, gold)
    for lo in range(0, 40000, 10000):
        parts(pred[lo:lo + 10000], gold[lo:lo + 10000])
    assert whole.get_metric()["BLEU"] == pytest.approx(parts.get_metric()["BLEU"], abs=1e-12)
