"""Host-side autograd helpers of the seq2seq modules that need no device."""
import torch

from probnmn.modules.seq2seq_base import _Alias, _SplitColumns


def test_split_columns_backward_is_the_concatenation():
    g = torch.Generator().manual_seed(0)
    w = torch.randn(12, 10, generator=g, requires_grad=True)
    a, b = _SplitColumns.apply(w, 4)
    assert torch.equal(a, w[:, :4]) and torch.equal(b, w[:, 4:])
    da, db = torch.randn(12, 4, generator=g), torch.randn(12, 6, generator=g)
    (a * da).sum().backward(retain_graph=True)  # one half only: the other half's gradient is zero
    assert torch.equal(w.grad[:, :4], da) and not w.grad[:, 4:].any()
    w.grad = None
    ((a * da).sum() + (b * db).sum()).backward()
    w2 = w.detach().clone().requires_grad_(True)
    ((w2[:, :4] * da).sum() + (w2[:, 4:] * db).sum()).backward()
    assert torch.equal(w.grad, w2.grad)


def test_alias_passes_the_gradient_to_both_terms():
    a, b = torch.randn(5, requires_grad=True), torch.randn(5, requires_grad=True)
    out = _Alias.apply(a, b, (a + b).detach())
    out.backward(torch.arange(5.0))
    assert torch.equal(a.grad, torch.arange(5.0)) and torch.equal(b.grad, torch.arange(5.0))
