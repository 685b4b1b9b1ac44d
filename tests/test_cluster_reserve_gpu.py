"""Data-parallel safety valve (pnmn_cluster_reserve_cus, probnmn.parallel.dp_safe): with CUs kept out of the grids of
the multi-CU recurrent kernels, a batch whose row tiles no longer fit one launch runs as several launches over row
ranges -- same results as the full-chip launch (every accumulation keeps its operand order: DESIGN 4.2)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def reserve():
    from probnmn import _hip

    lib = _hip.lib()
    full = lib.pnmn_cluster_reserve_cus(0)
    yield lambda n: lib.pnmn_cluster_reserve_cus(n)
    lib.pnmn_cluster_reserve_cus(0)
    assert lib.pnmn_cluster_reserve_cus(-1) == full


@pytest.mark.parametrize("B,with_tokens", [(1024, False), (1000, True), (520, False)])
def test_lstm_layer_in_row_ranges_equals_one_launch(reserve, B, with_tokens):
    from probnmn import _hip
    from probnmn.modules.seq2seq_base import _LSTMLayerSeq

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B)
    T, Hd, V = 9, 256, 40
    w = (torch.randn(4 * Hd, Hd, generator=g) * 0.05).to(dev)
    dhs = torch.randn(B, T, Hd, generator=g).to(dev)
    if with_tokens:
        xp0 = (torch.randn(V, 4 * Hd, generator=g) * 0.5).to(dev)
        tokens = torch.randint(0, V, (B, T), generator=g).to(dev)
    else:
        xp0, tokens = (torch.randn(B, T, 4 * Hd, generator=g) * 0.5).to(dev), None

    def run():
        xp = xp0.clone().requires_grad_(True)
        hs = _LSTMLayerSeq.apply(xp, w, None, None, tokens)
        hs.backward(dhs)
        return hs.detach().clone(), xp.grad.clone()

    full_cus = reserve(0)
    hs0, dx0 = run()
    left = reserve(32)
    assert left == full_cus - 32
    # (the library now covers the batch with launches of at most 8 * (left / 32) tiles)
    assert _hip.lib().pnmn_lstm_seq_workspace_bytes(B, 1) > 0
    hs1, dx1 = run()
    torch.cuda.synchronize()
    torch.testing.assert_close(hs1, hs0, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(dx1, dx0, rtol=1e-5, atol=1e-5)


def test_seq2seq_model_with_reserved_cus_equals_full_chip(reserve):
    """Whole ProgramGenerator pass (encoder LSTM layers + the multi-CU attention decoder, teacher forced) with and
    without reserved CUs."""
    from probnmn.data.synthetic import synthetic_batch
    from probnmn.models import ProgramGenerator
    from probnmn.vocabulary import Vocabulary

    dev = torch.device("cuda:0")
    vocab = Vocabulary.clevr()
    torch.manual_seed(0)
    pg = ProgramGenerator(vocab).to(dev)
    b = synthetic_batch(vocab, 600, seed=5, with_image=False)
    q, p = b["question"].to(dev), b["program"].to(dev)

    def run():
        pg.zero_grad()
        loss = pg(q, p, decoding_strategy="sampling")["loss"]
        loss.mean().backward()
        return loss.detach().clone(), {n: t.grad.clone() for n, t in pg.named_parameters() if t.grad is not None}

    reserve(0)
    l0, g0 = run()
    reserve(32)
    l1, g1 = run()
    torch.cuda.synchronize()
    torch.testing.assert_close(l1, l0, rtol=1e-5, atol=1e-5)
    for n in g0:
        scale = float(g0[n].abs().max()) or 1.0
        assert float((g1[n] - g0[n]).abs().max()) <= 2e-5 * scale, n
