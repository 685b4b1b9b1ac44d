"""Glue kernels of the seq2seq passes (csrc/seqglue.hip) against the tensor-op formulations they replace
(AllenNLP 0.9.0's add_sentence_boundary_token_ids / get_text_field_mask / get_final_encoder_states as restated
in probnmn.modules.seq2seq_base and oracle/seq2seq_oracle.py; reference seq2seq_base.py:278-293 for the trim).
Integer outputs bit-exact; float outputs exact where no sum is reordered, 1e-6 relative where one is."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _tokens(B, T, V, seed, empty_row=True):
    g = torch.Generator().manual_seed(seed)
    out = torch.zeros(B, T, dtype=torch.long)
    lens = torch.randint(1, T + 1, (B,), generator=g)
    lens[0] = T
    if empty_row and B > 2:
        lens[1] = 0  # an all-padding row
    for i in range(B):
        out[i, : lens[i]] = torch.randint(4, V, (int(lens[i]),), generator=g)
    return out


@pytest.mark.parametrize("B,T", [(1, 1), (5, 7), (130, 45), (64, 100)])
def test_token_prep_matches_the_tensor_ops(B, T):
    from probnmn.modules.seq2seq_base import _TokenPrep, add_sentence_boundary_token_ids

    pad, bos, eos = 0, 2, 3
    tok = _tokens(B, T, 50, seed=B + T).to(DEV)
    ref = add_sentence_boundary_token_ids(tok, pad, bos, eos)
    full, _, _ = _TokenPrep.run(tok, pad, bos, eos, drop_first=False, want_mask=False)
    assert torch.equal(full, ref)
    src, fmask, last = _TokenPrep.run(tok, pad, bos, eos, drop_first=True, want_mask=True)
    assert torch.equal(src, ref[:, 1:])
    assert torch.equal(fmask, (ref[:, 1:] != pad).float())
    assert torch.equal(last.long(), (ref[:, 1:] != pad).sum(1) - 1)
    # a strided view (rows of a wider matrix) is read in place
    wide = torch.cat((tok, tok), 1)
    full2, _, _ = _TokenPrep.run(wide[:, :T], pad, bos, eos, drop_first=False, want_mask=False)
    assert torch.equal(full2, ref)


def _trim_ref(predictions, end):
    steps = predictions.size(1)
    is_end = predictions == end
    has_end = is_end.any(1, keepdim=True)
    first = is_end.float().argmax(1, keepdim=True)
    pos = torch.arange(steps, device=predictions.device).unsqueeze(0)
    keep = torch.where(has_end, (pos <= first) & (first > 0), torch.ones_like(is_end))
    return predictions * keep


@pytest.mark.parametrize("B,T", [(3, 1), (9, 27), (257, 40), (33, 130)])
def test_trim_predictions(B, T):
    from probnmn.models import ProgramGenerator
    from probnmn.vocabulary import Vocabulary

    pg = ProgramGenerator(Vocabulary.clevr())
    g = torch.Generator().manual_seed(T)
    raw = torch.randint(0, 12, (B, T), generator=g)  # plenty of @end@ (3), zeros and repeats
    raw[0] = 5  # no @end@
    if B > 1:
        raw[1, 0] = pg._end_index  # starts with @end@
    raw = raw.to(DEV)
    assert torch.equal(pg._trim_predictions(raw), _trim_ref(raw, pg._end_index))


@pytest.mark.parametrize("B,T,H", [(2, 1, 256), (37, 46, 256), (130, 28, 256), (4, 9, 64)])
def test_mask_and_last(B, T, H):
    from probnmn.modules.seq2seq_base import _MaskAndLast

    torch.manual_seed(B)
    hs = torch.randn(B, T, H, device=DEV, requires_grad=True)
    lens = torch.randint(0, T + 1, (B,))
    lens[0] = T
    fmask = (torch.arange(T).unsqueeze(0) < lens.unsqueeze(1)).float().to(DEV)
    last = (lens - 1).to(torch.int32).to(DEV)  # -1 for an empty row: indexes the final (zeroed) step
    enc, hl = _MaskAndLast.apply(hs, fmask, last)
    hs2 = hs.detach().clone().requires_grad_(True)
    enc_ref = hs2 * fmask.unsqueeze(-1)
    hl_ref = enc_ref[torch.arange(B, device=DEV), last.long()]
    assert torch.equal(enc, enc_ref) and torch.equal(hl, hl_ref)
    w1, w2 = torch.randn_like(enc), torch.randn_like(hl)
    ((enc * w1).sum() + (hl * w2).sum()).backward()
    ((enc_ref * w1).sum() + (hl_ref * w2).sum()).backward()
    assert torch.allclose(hs.grad, hs2.grad, rtol=0, atol=1e-6)
    # only one of the two outputs used
    hs3 = hs.detach().clone().requires_grad_(True)
    _MaskAndLast.apply(hs3, fmask, last)[1].sum().backward()
    ref = torch.zeros_like(hs3)
    ref[torch.arange(B, device=DEV), last.long()] = 1.0
    assert torch.equal(hs3.grad, ref * fmask.unsqueeze(-1))


@pytest.mark.parametrize("B,T,C,V", [(1, 1, 64, 5), (130, 46, 1024, 96), (1024, 28, 1024, 44), (7, 30, 256, 128), (3, 70, 68, 9),
                                     (600, 1, 2052, 3)])
def test_embedding_grad(B, T, C, V):
    from probnmn.modules.seq2seq_base import embedding_grad

    g = torch.Generator().manual_seed(C + V)
    tok = torch.randint(0, V, (B, T), generator=g).to(DEV)
    dy = torch.randn(B, T, C, generator=g).to(DEV)
    onehot = torch.zeros(B * T, V, device=DEV, dtype=torch.float64)
    onehot.scatter_(1, tok.reshape(-1, 1), 1.0)
    ref = onehot.t() @ dy.reshape(B * T, C).double()
    out = embedding_grad(dy, tok, V)
    scale = float(ref.abs().max()) + 1e-9
    assert float((out.double() - ref).abs().max()) / scale < 2e-6
    # a skipped (padding) token, and the one-step shift of a free-running decoder's inputs
    ref_skip = ref.clone()
    ref_skip[1 % V] = 0
    out = embedding_grad(dy, tok, V, skip=1 % V)
    assert float((out.double() - ref_skip).abs().max()) / scale < 2e-6
    start = 2 % V
    tok_in = torch.cat((tok.new_full((B, 1), start), tok[:, :-1]), 1)
    onehot.zero_().scatter_(1, tok_in.reshape(-1, 1), 1.0)
    ref_shift = onehot.t() @ dy.reshape(B * T, C).double()
    out = embedding_grad(dy, tok, V, shift=True, start=start)
    assert float((out.double() - ref_shift).abs().max()) / (float(ref_shift.abs().max()) + 1e-9) < 2e-6


def test_derived_params_follow_the_parameters():
    """Fragment-order copies and bias sums from one launch equal the per-tensor formulation, are reused while the
    parameters are unchanged and brought up to date (in place, by re-running the cached job list) after an in-place
    update or a fused optimiser step; parameters that moved to other storage get a new job list."""
    from probnmn.models import ProgramGenerator
    from probnmn.modules.seq2seq_base import pack_fragments
    from probnmn.optim import ClampAdam
    from probnmn.vocabulary import Vocabulary

    torch.manual_seed(0)
    pg = ProgramGenerator(Vocabulary.clevr()).to(DEV)
    lstm, cell = pg._encoder._module, pg._decoder_cell

    def check(d):
        for layer in range(2):
            w = getattr(lstm, "weight_hh_l%d" % layer).detach()
            assert torch.equal(d["l%d.hh" % layer].view(-1), pack_fragments(w).view(-1))
            assert torch.equal(d["l%d.hhT" % layer].view(-1), pack_fragments(w.t()).view(-1))
            b = getattr(lstm, "bias_ih_l%d" % layer) + getattr(lstm, "bias_hh_l%d" % layer)
            assert torch.equal(d["l%d.b" % layer], b.detach())
        w_c = cell.weight_ih[:, :256].detach()
        assert torch.equal(d["d.c"].view(-1), pack_fragments(w_c).view(-1))
        assert torch.equal(d["d.cT"].view(-1), pack_fragments(w_c.t()).view(-1))
        assert torch.equal(d["d.hh"].view(-1), pack_fragments(cell.weight_hh.detach()).view(-1))
        assert torch.equal(d["d.hhT"].view(-1), pack_fragments(cell.weight_hh.detach().t()).view(-1))
        assert torch.equal(d["d.b"], (cell.bias_ih + cell.bias_hh).detach())

    d1 = pg._derived()
    check(d1)
    assert pg._derived() is d1  # cached
    with torch.no_grad():
        cell.weight_hh.mul_(1.5)  # an in-place update bumps the version counter
    d2 = pg._derived()
    check(d2)
    opt = ClampAdam(list(pg.parameters()), lr=1e-2)
    for p in pg.parameters():
        p.grad = torch.ones_like(p)
    opt.step()  # the fused step writes through pointers: the epoch counter invalidates the cache
    d3 = pg._derived()
    check(d3)
    assert pg._derived() is d3
    with torch.no_grad():  # new storage: the cached job list no longer describes the work
        cell.weight_hh.data = cell.weight_hh.data.clone() * 0.5
    check(pg._derived())


@pytest.mark.parametrize("V,K,N,sliced,pad", [(93, 256, 1024, False, 0), (44, 256, 1024, True, None), (128, 64, 128, False, 3), (5, 16, 64, True, None)])
def test_token_table_matches_linear(V, K, N, sliced, pad):
    """pnmn_token_table_fwd / _bwd against F.linear(embedding.weight, W, b) and autograd (weight as a column slice of
    a wider matrix, as the decoder cell's embedding half is; padding row without gradient)."""
    from probnmn.modules.seq2seq_base import _TokenTable

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(V + K)
    emb = torch.randn(V, K, generator=g).to(dev)
    if pad is not None:
        emb[pad] = 0.0
    wide = (torch.randn(N, 2 * K, generator=g) * 0.1).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    dtable = torch.randn(V, N, generator=g).to(dev)
    e1, w1, b1 = emb.clone().requires_grad_(True), wide.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    ref = torch.nn.functional.linear(e1, w1[:, K:] if sliced else w1[:, :K].contiguous(), b1)
    ref.backward(dtable)
    e2, w2, b2 = emb.clone().requires_grad_(True), wide.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    weight = w2[:, K:] if sliced else w2[:, :K].contiguous()
    got = _TokenTable.apply(e2, weight, b2, pad)
    got.backward(dtable)
    tol = dict(rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(got, ref, **tol)
    torch.testing.assert_close(b2.grad, b1.grad, rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(w2.grad, w1.grad, rtol=1e-5, atol=1e-4)
    want = e1.grad.clone()
    if pad is not None:
        want[pad] = 0.0
    torch.testing.assert_close(e2.grad, want, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("B,T,S", [(5, 26, 46), (130, 46, 27), (3, 1, 1), (17, 64, 64)])
def test_attn_denc_equals_the_two_batched_products(B, T, S):
    """pnmn_attn_denc: denc = w^T dctx + dscore^T h_prev (h_prev = [h0, hs[:, :-1]]) against torch.baddbmm."""
    from probnmn import _hip

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * 100 + T)
    r = lambda *shape: torch.randn(*shape, generator=g).to(dev)  # noqa: E731
    w, ds, dctx, hs, h0 = r(B, T, S).abs(), r(B, T, S), r(B, T, 256), r(B, T, 256), r(B, 256)
    denc = torch.full((B, S, 256), float("nan"), device=dev)
    _hip.check(_hip.lib().pnmn_attn_denc(w.data_ptr(), ds.data_ptr(), dctx.data_ptr(), hs.data_ptr(), h0.data_ptr(),
                                         denc.data_ptr(), B, T, S, 256, _hip.stream_ptr(dev)), "attn_denc")
    hprev = torch.cat((h0.unsqueeze(1), hs[:, :-1]), 1)
    want = torch.baddbmm(torch.bmm(w.transpose(1, 2), dctx), ds.transpose(1, 2), hprev)
    torch.testing.assert_close(denc, want, rtol=1e-4, atol=1e-4)
