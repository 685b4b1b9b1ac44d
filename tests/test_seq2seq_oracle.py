"""Pins for the seq2seq oracle (CPU).  AllenNLP 0.9.0 is absent, so the restated pieces are checked
against the torch primitives the real code sits on and against hand-derived known answers."""
import math

import torch
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

from oracle import detgen, seq2seq_oracle as so


def _sd(shapes, seed):
    gen = detgen.rng(seed)
    return {k: detgen.uniform(gen, v, 1.0 / math.sqrt(256)) for k, v in shapes.items()}


def test_packed_lstm_equals_nn_lstm_with_packing():
    torch.manual_seed(0)
    B, T, D, Hd = 5, 7, 12, 16
    lstm = torch.nn.LSTM(D, Hd, 2, batch_first=True)
    sd = {"enc." + n: p.detach().clone().requires_grad_(True) for n, p in lstm.named_parameters()}
    x = torch.randn(B, T, D)
    lengths = torch.tensor([7, 3, 1, 5, 2])
    mask = (torch.arange(T)[None, :] < lengths[:, None]).long()
    out = so.packed_lstm(sd, "enc.", x, mask)
    packed = pack_padded_sequence(x, lengths, batch_first=True, enforce_sorted=False)
    ref, _ = pad_packed_sequence(lstm(packed)[0], batch_first=True, total_length=T)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-6)
    assert float(out[1, 3:].abs().max()) == 0.0
    w = torch.randn(out.shape)
    (out * w).sum().backward()
    (ref * w).sum().backward()
    for n, p in lstm.named_parameters():
        torch.testing.assert_close(sd["enc." + n].grad, p.grad, rtol=1e-4, atol=1e-6)


def test_lstm_cell_equals_nn_lstmcell():
    torch.manual_seed(1)
    cell = torch.nn.LSTMCell(20, 8)
    x, h, c = torch.randn(3, 20), torch.randn(3, 8), torch.randn(3, 8)
    h2, c2 = so.lstm_cell(x, h, c, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh)
    rh, rc = cell(x, (h, c))
    torch.testing.assert_close(h2, rh)
    torch.testing.assert_close(c2, rc)


def test_boundary_tokens_and_trim_known_answers():
    t = torch.tensor([[5, 6, 7, 0], [8, 0, 0, 0], [4, 4, 4, 4]])
    out, mask = so.add_sentence_boundary_token_ids(t, t != 0, 2, 3)
    assert out.tolist() == [[2, 5, 6, 7, 3, 0], [2, 8, 3, 0, 0, 0], [2, 4, 4, 4, 4, 3]]
    assert mask.tolist() == [[1, 1, 1, 1, 1, 0], [1, 1, 1, 0, 0, 0], [1] * 6]
    p = torch.tensor([[9, 8, 3, 7, 3], [3, 9, 9, 9, 9], [9, 9, 9, 9, 9], [9, 3, 3, 3, 3]])
    assert so.trim_predictions(p).tolist() == [[9, 8, 3, 0, 0], [0] * 5, [9] * 5, [9, 3, 0, 0, 0]]


def test_masked_softmax_and_sequence_ce_known_answers():
    v = torch.tensor([[1.0, 2.0, 3.0, 4.0]])
    m = torch.tensor([[1, 1, 0, 0]])
    w = so.masked_softmax(v, m)
    # softmax over (1,2,0,0), masked and renormalised
    e = torch.exp(torch.tensor([1.0, 2.0, 0.0, 0.0]))
    p = e / e.sum()
    want = torch.tensor([[p[0], p[1], 0.0, 0.0]]) / (p[0] + p[1] + 1e-13)
    torch.testing.assert_close(w, want)
    logits = torch.zeros(1, 3, 4)
    ce = so.sequence_cross_entropy_with_logits(logits, torch.tensor([[1, 2, 0]]), torch.tensor([[1, 1, 0]]))
    torch.testing.assert_close(ce, torch.tensor([math.log(4.0)]))
    ce0 = so.sequence_cross_entropy_with_logits(logits, torch.tensor([[1, 2, 0]]), torch.tensor([[0, 0, 0]]))
    assert float(ce0) == 0.0  # empty mask -> 0 / 1e-13


def test_seq2seq_forward_shapes_and_teacher_forcing_consistency():
    vq, vp = 30, 44
    sd = _sd(so.seq2seq_param_shapes(vq, vp), 3)
    sd["_source_embedder.token_embedder_tokens.weight"][0] = 0
    src = torch.tensor([[5, 6, 7, 8, 0, 0], [9, 4, 0, 0, 0, 0], [4, 5, 6, 7, 8, 9]])
    tgt = torch.tensor([[10, 11, 12, 0], [13, 0, 0, 0], [14, 15, 16, 17]])
    out = so.seq2seq_forward(sd, src, tgt, "greedy")
    assert out["predictions"].shape == (3, 5) and out["loss"].shape == (3,)
    assert out["logits"].shape == (3, 5, vp)
    # the CE of row 1 only counts its 2 real targets (13, @end@)
    lp = F.log_softmax(out["logits"][1], -1)
    want = -(lp[0, 13] + lp[1, 3]) / 2
    torch.testing.assert_close(out["loss"][1], want, rtol=1e-5, atol=1e-6)
    # free-running greedy decode, then replaying its predictions through `forced_predictions`
    free = so.seq2seq_forward(sd, src, None, "greedy", max_decoding_steps=6)
    replay = so.seq2seq_forward(sd, src, None, "sampling", max_decoding_steps=6, forced_predictions=free["raw_predictions"])
    torch.testing.assert_close(free["loss"], replay["loss"])
    assert torch.equal(free["predictions"], replay["predictions"])
    # sampling never draws PAD / UNK / START
    g = torch.Generator().manual_seed(0)
    samp = so.seq2seq_forward(sd, src.repeat(20, 1), None, "sampling", max_decoding_steps=6, generator=g)
    assert not torch.isin(samp["raw_predictions"], torch.tensor([0, 1, 2])).any()
    # loss = - mean over kept steps of the step log-probs
    pm = (samp["predictions"] != 0).float()
    want = -(samp["step_logprobs"] * pm).sum(1) / (pm.sum(1) + 1e-12)
    torch.testing.assert_close(samp["loss"], want)


def test_program_prior_loss_ties_output_to_embedding():
    sd = _sd(so.prior_param_shapes(44), 4)
    progs = torch.tensor([[5, 6, 7, 0, 0], [8, 9, 10, 11, 12]])
    loss = so.program_prior_loss(sd, progs)
    assert loss.shape == (2,) and torch.isfinite(loss).all()
    w = sd["_embedder.token_embedder_programs.weight"].clone().requires_grad_(True)
    sd2 = dict(sd)
    sd2["_embedder.token_embedder_programs.weight"] = w
    so.program_prior_loss(sd2, progs).sum().backward()
    # gradient reaches the embedding both as input lookup and as output layer: rows of tokens that
    # never appear as inputs still get gradient through the tied softmax
    assert float(w.grad[40].abs().sum()) > 0


# ---- second, independent restatement (oracle/seq2seq_modules.py) against the first -------------------
def _ragged(B, T, V, seed, min_len=1):
    g = torch.Generator().manual_seed(seed)
    out = torch.zeros(B, T, dtype=torch.long)
    lens = torch.randint(min_len, T + 1, (B,), generator=g)
    lens[0], lens[1] = T, min_len
    for i in range(B):
        out[i, : lens[i]] = torch.randint(4, V, (int(lens[i]),), generator=g)
    return out


def test_two_independent_restatements_agree_whole_model():
    """Losses, predictions and every parameter gradient of the functional oracle (seq2seq_oracle.py) equal
    those of the nn.Module restatement (seq2seq_modules.py: nn.Embedding + packed nn.LSTM + nn.LSTMCell),
    teacher-forced, replayed samples, free-running greedy, and the program prior."""
    from oracle import seq2seq_modules as sm

    for v_src, v_tgt, t_src, t_tgt, seed in ((100, 44, 19, 13, 0), (44, 100, 13, 19, 1)):  # generator-like, reconstructor-like
        sd = _sd(so.seq2seq_param_shapes(v_src, v_tgt), 40 + seed)
        net = sm.ModuleSeq2Seq(v_src, v_tgt, max_decoding_steps=9)
        missing = net.load_state_dict(sd, strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
        src, tgt = _ragged(9, t_src, v_src, 50 + seed), _ragged(9, t_tgt, v_tgt, 60 + seed)

        def both(target, strategy, forced=None):
            leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
            a = so.seq2seq_forward(leaves, src, target, strategy, max_decoding_steps=9, forced_predictions=forced)
            a["loss"].mean().backward()
            net.zero_grad()
            b = net(src, target, strategy, forced)
            b["loss"].mean().backward()
            assert torch.equal(a["predictions"], b["predictions"])
            torch.testing.assert_close(a["loss"].detach(), b["loss"].detach(), rtol=1e-5, atol=1e-6)
            for name, p in net.named_parameters():
                torch.testing.assert_close(p.grad, leaves[name].grad, rtol=1e-4, atol=1e-6, msg=lambda m: name + ": " + m)
            return a

        both(tgt, "greedy")                                   # teacher forced: CE loss
        g = torch.Generator().manual_seed(70 + seed)
        forced = torch.randint(3, v_tgt, (9, 9), generator=g)  # replayed "samples", @end@ (3) included
        forced[0, 0] = 3                                       # a row whose first token is @end@ -> all padding
        forced[1] = 7                                          # a row that never ends
        both(None, "sampling", forced)
        free = both(None, "greedy")                           # free-running arg-max decode
        assert free["predictions"].shape == (9, 9)

    vocab = 44
    sd = _sd(so.prior_param_shapes(vocab), 80)
    prior = sm.ModulePrior(vocab)
    assert not prior.load_state_dict(sd, strict=True).missing_keys
    progs = _ragged(7, 11, vocab, 81)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    a = so.program_prior_loss(leaves, progs)
    a.mean().backward()
    b = prior(progs)
    b.mean().backward()
    torch.testing.assert_close(a.detach(), b.detach(), rtol=1e-5, atol=1e-6)
    for name, p in prior.named_parameters():
        torch.testing.assert_close(p.grad, leaves[name].grad, rtol=1e-4, atol=1e-6, msg=lambda m: name + ": " + m)
