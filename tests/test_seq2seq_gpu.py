"""Seq2seq models on the MI355X against the CPU oracle (same weights, same tokens).  fp32
tolerance: losses 1e-4, gradients 2e-3 of each tensor's max (BPTT over ~47 steps)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _tokens(B, T, V, seed, min_len=2):
    g = torch.Generator().manual_seed(seed)
    out = torch.zeros(B, T, dtype=torch.long)
    lens = torch.randint(min_len, T + 1, (B,), generator=g)
    lens[0] = T  # one full-length row
    for i in range(B):
        out[i, : lens[i]] = torch.randint(4, V, (int(lens[i]),), generator=g)
    return out


def _models(seed=0):
    from probnmn.models import ProgramGenerator, ProgramPrior, QuestionReconstructor
    from probnmn.vocabulary import Vocabulary

    vocab = Vocabulary.clevr()
    torch.manual_seed(seed)
    return vocab, ProgramGenerator(vocab), QuestionReconstructor(vocab), ProgramPrior(vocab, hidden_size=256)


def _cmp_grads(model, ref_sd, tag, tol=2e-3):
    worst = 0.0
    for name, p in model.named_parameters():
        g_ref = ref_sd[name].grad
        if g_ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        scale = float(g_ref.abs().max()) + 1e-12
        err = float((p.grad.cpu() - g_ref).abs().max()) / scale
        worst = max(worst, err)
        assert err < tol, (tag, name, err)
    return worst


def test_lstm_cell_kernel_matches_torch():
    from probnmn.modules.seq2seq_base import lstm_cell_pointwise

    g = torch.Generator().manual_seed(0)
    B, Hd = 37, 256
    gates = (torch.randn(B, 4 * Hd, generator=g) * 2).requires_grad_(True)
    c0 = torch.randn(B, Hd, generator=g).requires_grad_(True)
    i, f, gg, o = gates.chunk(4, 1)
    c_ref = torch.sigmoid(f) * c0 + torch.sigmoid(i) * torch.tanh(gg)
    h_ref = torch.sigmoid(o) * torch.tanh(c_ref)
    dh, dc = torch.randn(B, Hd, generator=g), torch.randn(B, Hd, generator=g)
    (h_ref * dh + c_ref * dc).sum().backward()
    gd = gates.detach().to(DEV).requires_grad_(True)
    cd = c0.detach().to(DEV).requires_grad_(True)
    h, c = lstm_cell_pointwise(gd, cd)
    (h * dh.to(DEV) + c * dc.to(DEV)).sum().backward()
    torch.testing.assert_close(h.detach().cpu(), h_ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(c.detach().cpu(), c_ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(gd.grad.cpu(), gates.grad, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(cd.grad.cpu(), c0.grad, rtol=1e-4, atol=1e-6)


def test_sampler_kernel_distribution_and_greedy():
    from probnmn.modules.seq2seq_base import choose_tokens

    g = torch.Generator().manual_seed(1)
    V = 100
    base = torch.randn(V, generator=g) * 2
    logits = base.unsqueeze(0).repeat(200000, 1).to(DEV)
    tok, lp = choose_tokens(logits, False, seed=1234, row_offset=0, step=3, pad=0, unk=1, start=2)
    tok, lp = tok.cpu(), lp.cpu()
    assert not torch.isin(tok, torch.tensor([0, 1, 2])).any()
    p = F.softmax(base, 0).clone()
    p[:3] = 0
    p = p / p.sum()
    freq = torch.bincount(tok, minlength=V).float() / tok.numel()
    # binomial std of a frequency ~ sqrt(p/N): allow 6 sigma + tiny absolute
    assert torch.all((freq - p).abs() < 6 * torch.sqrt(p / tok.numel()) + 1e-5)
    torch.testing.assert_close(lp, F.log_softmax(base, 0)[tok], rtol=1e-5, atol=1e-5)
    # same (seed,row,step) -> same draw; different step -> different stream
    tok2, _ = choose_tokens(logits, False, seed=1234, row_offset=0, step=3, pad=0, unk=1, start=2)
    tok3, _ = choose_tokens(logits, False, seed=1234, row_offset=0, step=4, pad=0, unk=1, start=2)
    assert torch.equal(tok2.cpu(), tok) and not torch.equal(tok3.cpu(), tok)
    # shard invariance: rows [1000, 2000) with offset 0 == rows [0, 1000) with offset 1000
    a, _ = choose_tokens(logits[:2000], False, 99, 0, 0, 0, 1, 2)
    b, _ = choose_tokens(logits[:1000], False, 99, 1000, 0, 0, 1, 2)
    assert torch.equal(a[1000:], b)
    gt, glp = choose_tokens(torch.randn(64, 44, generator=g).to(DEV), True, 0, 0, 0, 0, 1, 2)
    assert gt.shape == (64,)


@pytest.mark.parametrize("which", ["pg", "qr"])
def test_teacher_forced_and_sampled_match_oracle(which):
    from oracle import seq2seq_oracle as so

    vocab, pg, qr, _ = _models()
    model = pg if which == "pg" else qr
    vq, vp = vocab.get_vocab_size("questions"), vocab.get_vocab_size("programs")
    B = 12
    src = _tokens(B, 20 if which == "pg" else 12, vq if which == "pg" else vp, 5)
    tgt = _tokens(B, 12 if which == "pg" else 20, vp if which == "pg" else vq, 6)
    cpu_sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(DEV).train()

    # 1. teacher forcing: CE loss and its gradients
    ref_sd = {k: v.clone().requires_grad_(True) for k, v in cpu_sd.items()}
    ref = so.seq2seq_forward(ref_sd, src, tgt, "greedy")
    ref["loss"].mean().backward()
    out = model(src.to(DEV), tgt.to(DEV), decoding_strategy="greedy")
    out["loss"].mean().backward()
    torch.testing.assert_close(out["loss"].detach().cpu(), ref["loss"].detach(), rtol=1e-4, atol=1e-4)
    assert torch.equal(out["predictions"].cpu(), ref["predictions"])
    print(which, "teacher-forced worst grad err", _cmp_grads(model, ref_sd, which + "/tf"))

    # 2. sampling: replay the device's samples through the oracle
    model.zero_grad(set_to_none=True)
    torch.manual_seed(7)
    out = model(src.to(DEV), None, decoding_strategy="sampling")
    out["loss"].mean().backward()
    pred = out["predictions"].cpu()
    assert pred.shape == (B, model._max_decoding_steps)
    assert not torch.isin(pred, torch.tensor([1, 2])).any()
    ref_sd = {k: v.clone().requires_grad_(True) for k, v in cpu_sd.items()}
    ref = so.seq2seq_forward(ref_sd, src, None, "sampling", max_decoding_steps=model._max_decoding_steps,
                             forced_predictions=pred)
    ref["loss"].mean().backward()
    assert torch.equal(ref["predictions"], pred)
    torch.testing.assert_close(out["loss"].detach().cpu(), ref["loss"].detach(), rtol=1e-4, atol=1e-4)
    print(which, "sampled worst grad err", _cmp_grads(model, ref_sd, which + "/sample"))

    # 3. free-running greedy in eval mode
    model.eval()
    with torch.no_grad():
        out = model(src.to(DEV), None, decoding_strategy="greedy")
    ref = so.seq2seq_forward(cpu_sd, src, None, "greedy", max_decoding_steps=model._max_decoding_steps)
    same = (out["predictions"].cpu() == ref["predictions"]).all(1).float().mean()
    assert same >= 0.9  # an arg-max near-tie may diverge a row; the rest must be identical
    # 4. eval with targets records metrics
    model(src.to(DEV), tgt.to(DEV), decoding_strategy="greedy")
    m = model.get_metrics()
    assert set(m) == {"BLEU", "perplexity", "sequence_accuracy", "word_error_rate"} and 0.0 <= m["BLEU"] <= 1.0


def test_trim_predictions_matches_reference_rule():
    from oracle import seq2seq_oracle as so

    _, pg, _, _ = _models()
    p = torch.tensor([[9, 8, 3, 7, 3], [3, 9, 9, 9, 9], [9, 9, 9, 9, 9], [9, 3, 3, 3, 3], [0, 0, 3, 0, 0]])
    assert torch.equal(pg._trim_predictions(p.to(DEV)).cpu(), so.trim_predictions(p))


def test_program_prior_loss_and_gradients():
    from oracle import seq2seq_oracle as so

    vocab, _, _, prior = _models()
    progs = _tokens(10, 26, 44, 9)
    cpu_sd = {k: v.detach().clone() for k, v in prior.state_dict().items() if k != "_output_layer.weight"}
    ref_sd = {k: v.clone().requires_grad_(True) for k, v in cpu_sd.items()}
    ref = so.program_prior_loss(ref_sd, progs)
    ref.mean().backward()
    prior.to(DEV).eval()
    out = prior(progs.to(DEV))
    out["loss"].mean().backward()
    torch.testing.assert_close(out["loss"].detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-4)
    assert out["predictions"].shape == (10, 27)
    for name, p in prior.named_parameters():
        g_ref = ref_sd[name].grad
        scale = float(g_ref.abs().max()) + 1e-12
        assert float((p.grad.cpu() - g_ref).abs().max()) / scale < 2e-3, name


def test_persistent_lstm_layer_matches_nn_lstm():
    """pnmn_lstm_seq_fwd/bwd against torch.nn.LSTM (one layer, hidden 256), ragged batch size."""
    from probnmn.modules.seq2seq_base import _LSTMLayerSeq

    torch.manual_seed(3)
    B, T, D, Hd = 37, 11, 256, 256
    lstm = torch.nn.LSTM(D, Hd, 1, batch_first=True)
    x = torch.randn(B, T, D)
    ref, _ = lstm(x)
    w = torch.randn(ref.shape)
    (ref * w).sum().backward()
    xd = x.to(DEV).requires_grad_(True)
    w_ih = lstm.weight_ih_l0.detach().to(DEV).requires_grad_(True)
    w_hh = lstm.weight_hh_l0.detach().to(DEV).requires_grad_(True)
    bias = (lstm.bias_ih_l0 + lstm.bias_hh_l0).detach().to(DEV).requires_grad_(True)
    out = _LSTMLayerSeq.apply(F.linear(xd, w_ih, bias), w_hh)
    (out * w.to(DEV)).sum().backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(w_hh.grad.cpu(), lstm.weight_hh_l0.grad, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(w_ih.grad.cpu(), lstm.weight_ih_l0.grad, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(bias.grad.cpu(), lstm.bias_ih_l0.grad, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("multi_cu", ["0", "1"])
@pytest.mark.parametrize("degenerate", [False, True])
def test_persistent_decoder_matches_stepwise_torch(degenerate, multi_cu, monkeypatch):
    """pnmn_attn_lstm_fwd/bwd (teacher forced) against the same recurrence written with torch ops on
    the device: hidden states and every gradient (xe, enc, h0, W_c, W_hh)."""
    from probnmn.modules.seq2seq_base import _AttnLSTMDecoder, masked_softmax

    monkeypatch.setenv("PNMN_DECODER_CLUSTER", multi_cu)
    torch.manual_seed(0)
    B, T, S, Hd, V = 19, 7, 11, 256, 44
    enc = torch.randn(B, S, Hd, device=DEV).requires_grad_(True)
    lens = torch.randint(1, S + 1, (B,), device=DEV)
    lens[0] = S
    if degenerate:
        lens[:] = 1
    mask = (torch.arange(S, device=DEV)[None, :] < lens[:, None]).float()
    h0 = torch.randn(B, Hd, device=DEV).requires_grad_(True)
    w_c = (torch.randn(4 * Hd, Hd, device=DEV) * 0.05).requires_grad_(True)
    w_hh = (torch.randn(4 * Hd, Hd, device=DEV) * 0.05).requires_grad_(True)
    xe = torch.randn(B, T, 4 * Hd, device=DEV).requires_grad_(True)
    w_p, b_p = torch.randn(V, Hd, device=DEV), torch.randn(V, device=DEV)
    wgt = torch.randn(B, T, Hd, device=DEV)
    leaves = dict(xe=xe, enc=enc, h0=h0, w_c=w_c, w_hh=w_hh)
    hs, _ = _AttnLSTMDecoder.apply(xe, None, enc, mask, h0, w_c, w_hh, w_p, b_p, 0, T, 1, 0, 0, 1, 2)
    (hs * wgt).sum().backward()
    got = {k: v.grad.clone() for k, v in leaves.items()}
    for v in leaves.values():
        v.grad = None
    h, c, outs = h0, torch.zeros_like(h0), []
    for t in range(T):
        w = masked_softmax(torch.bmm(enc, h.unsqueeze(-1)).squeeze(-1), mask)
        ctx = torch.bmm(w.unsqueeze(1), enc).squeeze(1)
        i, f, g, o = (xe[:, t] + ctx @ w_c.t() + h @ w_hh.t()).chunk(4, 1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs.append(h)
    ref = torch.stack(outs, 1)
    torch.testing.assert_close(hs.detach(), ref.detach(), rtol=1e-4, atol=1e-5)
    (ref * wgt).sum().backward()
    for k, v in leaves.items():
        scale = float(v.grad.abs().max())
        assert float((v.grad - got[k]).abs().max()) / scale < 1e-4, k


def test_decoder_sampling_is_reproducible_and_shard_invariant():
    """Free-running sampling inside the persistent kernel: same seed -> same programs; a shard with a
    row offset draws what the full batch drew for those rows."""
    vocab, pg, _, _ = _models()
    pg.to(DEV).eval()
    src = _tokens(40, 20, vocab.get_vocab_size("questions"), 5).to(DEV)
    with torch.no_grad():
        torch.manual_seed(11)
        a = pg(src, None, "sampling")["predictions"]
        torch.manual_seed(11)
        b = pg(src, None, "sampling")["predictions"]
        torch.manual_seed(11)
        pg.sample_row_offset = 16
        c = pg(src[16:], None, "sampling")["predictions"]
        pg.sample_row_offset = 0
    assert torch.equal(a, b)
    assert torch.equal(a[16:], c)
    assert not torch.isin(a, torch.tensor([1, 2], device=DEV)).any()


@pytest.mark.parametrize("batch", [5, 16, 100, 128, 512, 1000, 1024])
def test_multi_cu_lstm_layer_matches_one_workgroup_per_tile(batch, monkeypatch):
    """pnmn_lstm_seq_fwd/_bwd with a workspace (4 or 8 workgroups per 16-row tile, W_hh slices in
    registers, recurrent vector exchanged through L2) against the same entry points without one."""
    from probnmn.modules.seq2seq_base import _LSTMLayerSeq

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(batch)
    T, H = 23, 256
    xp = (torch.randn(batch, T, 4 * H, generator=g) * 0.7).to(dev)
    w = (torch.randn(4 * H, H, generator=g) * 0.06).to(dev)
    dhs = torch.randn(batch, T, H, generator=g).to(dev)

    def run():
        x, ww = xp.clone().requires_grad_(True), w.clone().requires_grad_(True)
        hs = _LSTMLayerSeq.apply(x, ww)
        hs.backward(dhs)
        return hs.detach(), x.grad, ww.grad

    monkeypatch.setenv("PNMN_LSTM_CLUSTER", "0")
    ref = run()
    monkeypatch.setenv("PNMN_LSTM_CLUSTER", "1")
    first = None
    for _ in range(3):
        got = run()
        torch.cuda.synchronize()
        # same operand order in every accumulation; only the compiler's fma contraction of the cell
        # update may differ between the two kernels
        for a, b in zip(got, ref):
            assert float((a - b).abs().max()) < 2e-5 * max(1.0, float(b.abs().max()))
        if first is None:
            first = got
        else:  # hand-off races would show up as run-to-run differences
            assert torch.equal(got[0], first[0]) and torch.equal(got[1], first[1])


@pytest.mark.parametrize("batch,cluster", [(5, "1"), (128, "1"), (1000, "1"), (130, "0")])
def test_lstm_layer_reads_its_input_projection_from_the_token_table(batch, cluster, monkeypatch):
    """pnmn_lstm_seq_fwd with `tokens`: row tokens[b, t] of the [V, 4H] table instead of xp[b, t] -- same states
    bit for bit as with F.embedding(tokens, table) written out, and the table's gradient = the scatter-add of
    the gate gradients (tokens as a strided view, as the models pass them)."""
    from probnmn.modules.seq2seq_base import _LSTMLayerSeq

    monkeypatch.setenv("PNMN_LSTM_CLUSTER", cluster)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(batch)
    T, H, V = 21, 256, 93
    table = (torch.randn(V, 4 * H, generator=g) * 0.7).to(dev)
    w = (torch.randn(4 * H, H, generator=g) * 0.06).to(dev)
    tokens = torch.randint(0, V, (batch, T + 3), generator=g).to(dev)[:, 1 : T + 1]
    dhs = torch.randn(batch, T, H, generator=g).to(dev)
    ta, wa = table.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ha = _LSTMLayerSeq.apply(F.embedding(tokens, ta), wa)
    ha.backward(dhs)
    tb, wb = table.clone().requires_grad_(True), w.clone().requires_grad_(True)
    hb = _LSTMLayerSeq.apply(tb, wb, None, None, tokens)
    hb.backward(dhs)
    assert torch.equal(ha, hb)
    assert torch.equal(wa.grad, wb.grad)
    torch.testing.assert_close(tb.grad, ta.grad, rtol=1e-5, atol=1e-5 * float(ta.grad.abs().max()))


@pytest.mark.parametrize("batch,multi", [(7, "1"), (128, "1"), (530, "1"), (40, "0")])
def test_decoder_teacher_forcing_from_the_token_table(batch, multi, monkeypatch):
    """pnmn_attn_lstm_fwd(_multi) with in_tokens: the teacher-forced inputs as rows of etable."""
    from probnmn.modules.seq2seq_base import _AttnLSTMDecoder

    monkeypatch.setenv("PNMN_DECODER_CLUSTER", multi)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(batch + 1)
    T, S, H, V = 12, 27, 256, 44
    etable = (torch.randn(V, 4 * H, generator=g) * 0.5).to(dev)
    enc = torch.randn(batch, S, H, generator=g).to(dev)
    mask = (torch.rand(batch, S, generator=g) < 0.8).float().to(dev)
    mask[:, 0] = 1.0
    h0 = torch.randn(batch, H, generator=g).to(dev)
    w_c = (torch.randn(4 * H, H, generator=g) * 0.05).to(dev)
    w_hh = (torch.randn(4 * H, H, generator=g) * 0.05).to(dev)
    tokens = torch.randint(0, V, (batch, T + 2), generator=g).to(dev)[:, :T]
    dhs = torch.randn(batch, T, H, generator=g).to(dev)
    outs = []
    for fused in (False, True):
        et, e, h = etable.clone().requires_grad_(True), enc.clone().requires_grad_(True), h0.clone().requires_grad_(True)
        if fused:
            hs, _ = _AttnLSTMDecoder.apply(None, et, e, mask, h, w_c, w_hh, None, None, 0, T, 5, 0, 0, 1, 2, None, tokens)
        else:
            hs, _ = _AttnLSTMDecoder.apply(F.embedding(tokens, et), None, e, mask, h, w_c, w_hh, None, None, 0, T, 5, 0, 0, 1, 2)
        hs.backward(dhs)
        outs.append((hs.detach(), e.grad, h.grad, et.grad))
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    torch.testing.assert_close(outs[1][3], outs[0][3], rtol=1e-5, atol=1e-5 * float(outs[0][3].abs().max()))


def test_multi_cu_hand_off_counters_per_stream_and_under_capture():
    """The hand-off counters live in a block the library keeps per stream and the kernels leave zeroed
    (cluster.h): launches on two streams at once must not share one, and a launch captured into a graph
    (counters in the caller's workspace, zeroed by a kernel node) must replay correctly -- also next to
    eager launches on the stream it was captured on."""
    from probnmn.modules.seq2seq_base import _LSTMLayerSeq

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(77)
    B, T, H = 128, 19, 256
    xs = [(torch.randn(B, T, 4 * H, generator=g) * 0.7).to(dev) for _ in range(2)]
    w = (torch.randn(4 * H, H, generator=g) * 0.06).to(dev)
    with torch.no_grad():
        ref = [_LSTMLayerSeq.apply(x, w).clone() for x in xs]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
        for _ in range(3):
            outs = []
            for x, st in zip(xs, streams):
                with torch.cuda.stream(st):
                    for _ in range(4):
                        out = _LSTMLayerSeq.apply(x, w)
                    outs.append(out)
            torch.cuda.synchronize()
            assert torch.equal(outs[0], ref[0]) and torch.equal(outs[1], ref[1])
        side = torch.cuda.Stream(dev)
        with torch.cuda.stream(side):
            _LSTMLayerSeq.apply(xs[0], w)  # allocations of the pass are in the pool before the capture
            side.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                captured = _LSTMLayerSeq.apply(xs[0], w)
            for _ in range(3):
                graph.replay()
                eager = _LSTMLayerSeq.apply(xs[1], w)
            side.synchronize()
        assert torch.equal(captured, ref[0]) and torch.equal(eager, ref[1])


@pytest.mark.parametrize("batch,steps,positions", [(7, 5, 3), (128, 27, 46), (530, 9, 27), (1024, 12, 64)])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_multi_cu_decoder_matches_one_workgroup_per_tile(batch, steps, positions, mode, monkeypatch):
    """pnmn_attn_lstm_fwd_multi / _bwd_multi (eight workgroups per tile, batches beyond 512 rows in
    several launches) against pnmn_attn_lstm_fwd / _bwd: hidden states, chosen tokens and every gradient."""
    from probnmn.modules.seq2seq_base import _AttnLSTMDecoder

    g = torch.Generator().manual_seed(batch + mode)
    B, T, S, Hd, V = batch, steps, positions, 256, 44
    r = lambda *shape, scale=1.0: (torch.randn(*shape, generator=g) * scale).to(DEV)  # noqa: E731
    enc0, h00 = r(B, S, Hd), r(B, Hd)
    lens = torch.randint(1, S + 1, (B,), generator=g).to(DEV)
    mask = (torch.arange(S, device=DEV)[None, :] < lens[:, None]).float()
    w_c0, w_hh0 = r(4 * Hd, Hd, scale=0.05), r(4 * Hd, Hd, scale=0.05)
    xe0, etable0 = r(B, T, 4 * Hd), r(V, 4 * Hd)
    w_p, b_p = r(V, Hd, scale=0.3), r(V)
    wgt = r(B, T, Hd)

    def run():
        leaves = dict(enc=enc0.clone().requires_grad_(True), h0=h00.clone().requires_grad_(True),
                      w_c=w_c0.clone().requires_grad_(True), w_hh=w_hh0.clone().requires_grad_(True))
        if mode == 0:
            leaves["xe"] = xe0.clone().requires_grad_(True)
            hs, tok = _AttnLSTMDecoder.apply(leaves["xe"], None, leaves["enc"], mask, leaves["h0"], leaves["w_c"],
                                             leaves["w_hh"], w_p, b_p, 0, T, 5, 3, 0, 1, 2)
        else:
            leaves["etable"] = etable0.clone().requires_grad_(True)
            hs, tok = _AttnLSTMDecoder.apply(None, leaves["etable"], leaves["enc"], mask, leaves["h0"], leaves["w_c"],
                                             leaves["w_hh"], w_p, b_p, mode, T, 5, 3, 0, 1, 2)
        (hs * wgt).sum().backward()
        torch.cuda.synchronize()
        return hs.detach(), tok, {k: v.grad for k, v in leaves.items()}

    monkeypatch.setenv("PNMN_DECODER_CLUSTER", "0")
    hs_ref, tok_ref, grads_ref = run()
    monkeypatch.setenv("PNMN_DECODER_CLUSTER", "1")
    hs, tok, grads = run()
    hs2, tok2, grads2 = run()
    assert torch.equal(hs, hs2) and torch.equal(tok, tok2)  # hand-off races would show up as run-to-run differences
    for k in grads:
        if k != "etable":  # (index_add_ with atomics: summation order varies)
            assert torch.equal(grads[k], grads2[k]), k
    if mode == 0:
        same = torch.ones(B, dtype=torch.bool, device=DEV)
    else:
        # a token can flip where two candidates are within round-off of the draw; such a row then diverges
        same = (tok == tok_ref).all(1)
        assert float(same.float().mean()) > 0.97
    assert float((hs - hs_ref)[same].abs().max()) < 2e-4
    if bool(same.all()):
        for k, ref in grads_ref.items():
            scale = float(ref.abs().max()) + 1e-12
            assert float((grads[k] - ref).abs().max()) / scale < 2e-4, k


def test_config5_forty_step_programs():
    """BASELINE config 5: programs of up to 40 tokens.  The generator decodes 40 steps in the fused
    persistent kernel (the reference hard-codes 26, program_generator.py:34-35: here a constructor
    keyword), the reconstructor encodes 40-token programs (42 source positions) and the prior scores
    them; each against the oracle on the same tokens."""
    from oracle import seq2seq_oracle as so
    from probnmn.models import ProgramGenerator, ProgramPrior, QuestionReconstructor
    from probnmn.vocabulary import Vocabulary

    vocab = Vocabulary.clevr()
    torch.manual_seed(2)
    pg = ProgramGenerator(vocab, max_decoding_steps=40)
    qr, prior = QuestionReconstructor(vocab), ProgramPrior(vocab, hidden_size=256)
    vq, vp = vocab.get_vocab_size("questions"), vocab.get_vocab_size("programs")
    B = 24
    questions = _tokens(B, 45, vq, 15)
    sds = {n: {k: v.detach().clone() for k, v in m.state_dict().items()} for n, m in (("pg", pg), ("qr", qr), ("prior", prior))}
    for m in (pg, qr, prior):
        m.to(DEV)
    pg.train()
    qr.train()
    prior.eval()

    # sampling decode, 40 steps, replayed through the oracle
    out = pg(questions.to(DEV), None, decoding_strategy="sampling")
    out["loss"].mean().backward()
    z = out["predictions"].cpu()
    assert z.shape == (B, 40)
    ref_sd = {k: v.clone().requires_grad_(True) for k, v in sds["pg"].items()}
    ref = so.seq2seq_forward(ref_sd, questions, None, "sampling", max_decoding_steps=40, forced_predictions=z)
    ref["loss"].mean().backward()
    assert torch.equal(ref["predictions"], z)
    torch.testing.assert_close(out["loss"].detach().cpu(), ref["loss"].detach(), rtol=1e-4, atol=1e-4)
    print("pg/40 worst grad err", _cmp_grads(pg, ref_sd, "pg/40"))

    # teacher forced on full-length 40-token programs
    programs = _tokens(B, 40, vp, 16, min_len=30)
    pg.zero_grad(set_to_none=True)
    out = pg(questions.to(DEV), programs.to(DEV), decoding_strategy="sampling")
    out["loss"].mean().backward()
    ref_sd = {k: v.clone().requires_grad_(True) for k, v in sds["pg"].items()}
    ref = so.seq2seq_forward(ref_sd, questions, programs, "greedy")
    ref["loss"].mean().backward()
    torch.testing.assert_close(out["loss"].detach().cpu(), ref["loss"].detach(), rtol=1e-4, atol=1e-4)
    print("pg/40 teacher-forced worst grad err", _cmp_grads(pg, ref_sd, "pg/40tf"))

    # reconstructor: 40-token programs as the source
    out = qr(programs.to(DEV), questions.to(DEV), decoding_strategy="sampling")
    out["loss"].mean().backward()
    ref_sd = {k: v.clone().requires_grad_(True) for k, v in sds["qr"].items()}
    ref = so.seq2seq_forward(ref_sd, programs, questions, "greedy")
    ref["loss"].mean().backward()
    torch.testing.assert_close(out["loss"].detach().cpu(), ref["loss"].detach(), rtol=1e-4, atol=1e-4)
    print("qr/40 worst grad err", _cmp_grads(qr, ref_sd, "qr/40"))

    # prior: -log p(z) of the 40-token programs
    with torch.no_grad():
        got = prior(programs.to(DEV))["loss"].cpu()
    psd = {k: v for k, v in sds["prior"].items() if k != "_output_layer.weight"}
    torch.testing.assert_close(got, so.program_prior_loss(psd, programs).detach(), rtol=1e-4, atol=1e-4)


def test_program_prior_sample_and_training_step():
    """ProgramPrior.sample (program_prior.py:174-301) against the oracle on forced draws, its free
    draws' invariants, and one ProgramPriorStep (program_prior_trainer.py:79-90) against the oracle's
    gradient (compared before Adam)."""
    from oracle import seq2seq_oracle as so
    from probnmn.trainers.module_training import ProgramPriorStep

    vocab, _, _, prior = _models(seed=4)
    sd = {k: v.detach().clone() for k, v in prior.state_dict().items() if k != "_output_layer.weight"}
    prior.to(DEV).eval()
    g = torch.Generator().manual_seed(12)
    forced = torch.randint(3, 44, (9, 27), generator=g)
    forced[0, 0] = 3   # starts with @end@ -> all padding
    forced[1] = 9      # never ends -> kept whole
    got = prior.sample(9, 28, _forced=forced)
    want = so.program_prior_sample(sd, forced, 28)
    assert torch.equal(got["predictions"].cpu(), want["predictions"])
    torch.testing.assert_close(got["loss"].cpu(), want["loss"], rtol=1e-4, atol=1e-5)
    free = prior.sample(64, 28)
    assert free["predictions"].shape == (64, 27) and not torch.isin(free["predictions"], torch.tensor([1, 2], device=DEV)).any()
    assert bool((free["loss"][1:] >= free["loss"][:-1] - 1e-6).all())  # most likely (smallest loss) first

    progs = _tokens(16, 26, 44, 13)
    step = ProgramPriorStep(prior, lr=1e-2)
    out = step.step({"program": progs.to(DEV)})
    ref_sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = so.program_prior_loss(ref_sd, progs).mean()
    ref.backward()
    assert float(out["loss"]) == pytest.approx(float(ref), rel=1e-4)
    for name, p in prior.named_parameters():
        want_g = ref_sd[name].grad
        assert float((p.grad.cpu() - want_g).abs().max()) / (float(want_g.abs().max()) + 1e-12) < 2e-3, name
    assert step.after_validation(1.0 / 3.0) == 1e-2 and set(step.state_dict()) == {"program_prior", "optimizer", "scheduler", "iteration"}


@pytest.mark.parametrize("rows_a,rows_b", [(40, 70), (16, 16), (300, 300)])
def test_paired_teacher_forced_decodes_equal_two_single_ones(rows_a, rows_b):
    """Seq2SeqBase.decode_prepare / decode_pair: the generator's and the reconstructor's teacher-forced decodes in one
    launch each way (attn_lstm_fwd_pair_kernel / bwd) against the same two passes launched one after the other -- same
    kernels' bodies, same operand order: losses and every gradient bit for bit.  (300 + 300 rows do not fit one launch
    together: the library then runs them back to back.)"""
    from probnmn.data.synthetic import synthetic_batch
    from probnmn.models import ProgramGenerator, QuestionReconstructor
    from probnmn.modules.seq2seq_base import decode_pair
    from probnmn.vocabulary import Vocabulary

    dev = torch.device("cuda:0")
    vocab = Vocabulary.clevr()
    torch.manual_seed(3)
    pg, qr = ProgramGenerator(vocab).to(dev), QuestionReconstructor(vocab).to(dev)
    ba = synthetic_batch(vocab, rows_a, seed=31, with_image=False)
    bb = synthetic_batch(vocab, rows_b, seed=32, with_image=False)
    qa, pa = ba["question"].to(dev), ba["program"].to(dev)
    qb, pb = bb["question"].to(dev), bb["program"].to(dev)
    wa, wb = torch.rand(rows_a, device=dev), torch.rand(rows_b, device=dev)  # row weights: every row's gradient matters

    def run(paired):
        for m in (pg, qr):
            m.train()
            m.zero_grad(set_to_none=True)
        sa, sb = pg.encode(qa), qr.encode(pb)
        if paired:
            prep_a, prep_b = pg.decode_prepare(sa, pa), qr.decode_prepare(sb, qb)
            assert prep_a is not None and prep_b is not None
            oa, ob = decode_pair(prep_a, prep_b)
        else:
            oa = pg.decode(sa, pa, "sampling", need_predictions=False)
            ob = qr.decode(sb, qb, "sampling", need_predictions=False)
        ((oa["loss"] * wa).sum() + (ob["loss"] * wb).sum()).backward()
        torch.cuda.synchronize()
        grads = {("pg", n): p.grad.clone() for n, p in pg.named_parameters()}
        grads.update({("qr", n): p.grad.clone() for n, p in qr.named_parameters()})
        return oa["loss"].detach().clone(), ob["loss"].detach().clone(), grads

    la0, lb0, g0 = run(False)
    la1, lb1, g1 = run(True)
    assert torch.equal(la0, la1) and torch.equal(lb0, lb1)
    for k in g0:
        # (split-K weight-gradient GEMMs and atomics-free kernels: same order in both runs)
        torch.testing.assert_close(g1[k], g0[k], rtol=1e-6, atol=1e-7, msg=lambda m, k=k: "%s: %s" % (k, m))


@pytest.mark.parametrize("rows_s,rows_t", [(48, 80), (256, 256), (400, 300)])
def test_paired_sampling_and_teacher_forced_decodes_equal_two_single_ones(rows_s, rows_t):
    """What the training iterations launch: the generator's SAMPLING decode of the unsupervised rows and its
    teacher-forced decode of the supervised rows as one launch each way, against the two passes one after the other
    (same seed): sampled programs, both losses and every gradient identical."""
    from probnmn.data.synthetic import synthetic_batch
    from probnmn.models import ProgramGenerator
    from probnmn.modules.seq2seq_base import decode_pair
    from probnmn.vocabulary import Vocabulary

    dev = torch.device("cuda:0")
    vocab = Vocabulary.clevr()
    torch.manual_seed(5)
    pg = ProgramGenerator(vocab).to(dev)
    pg.sample_row_offset = 1000
    b = synthetic_batch(vocab, rows_s + rows_t, seed=41, with_image=False)
    q, p = b["question"].to(dev), b["program"].to(dev)
    idx_s, idx_t = torch.arange(rows_s, device=dev), torch.arange(rows_s, rows_s + rows_t, device=dev)
    ws, wt = torch.rand(rows_s, device=dev), torch.rand(rows_t, device=dev)

    def run(paired):
        pg.train()
        pg.zero_grad(set_to_none=True)
        torch.manual_seed(77)  # the sampling seed comes from the CPU generator
        state = pg.encode(q)
        st_s, st_t = pg.select_rows(state, idx_s), pg.select_rows(state, idx_t)
        if paired:
            prep_s, prep_t = pg.decode_prepare(st_s, None, "sampling"), pg.decode_prepare(st_t, p[idx_t])
            assert prep_s is not None and prep_t is not None
            o_s, o_t = decode_pair(prep_s, prep_t)
        else:
            o_s = pg.decode(st_s, None, "sampling")
            o_t = pg.decode(st_t, p[idx_t], "sampling", need_predictions=False)
        after = int(torch.randint(0, 2 ** 62, (1,)).item())  # where the CPU generator stands behind the two passes
        ((o_s["loss"] * ws).sum() + (o_t["loss"] * wt).sum()).backward()
        torch.cuda.synchronize()
        return (o_s["predictions"].clone(), o_s["loss"].detach().clone(), o_t["loss"].detach().clone(),
                {n: t.grad.clone() for n, t in pg.named_parameters()}, after)

    z0, ls0, lt0, g0, after0 = run(False)
    z1, ls1, lt1, g1, after1 = run(True)
    assert torch.equal(z0, z1) and torch.equal(ls0, ls1) and torch.equal(lt0, lt1)
    assert after0 == after1  # (one seed draw per pass in either schedule -- the next iteration samples alike)
    assert int((z0 != 0).sum()) > rows_s  # real programs were sampled
    for n in g0:
        torch.testing.assert_close(g1[n], g0[n], rtol=1e-6, atol=1e-7, msg=lambda m, n=n: "%s: %s" % (n, m))


def test_pairing_falls_back_without_the_cluster_kernels(monkeypatch):
    """decode_prepare declines when the multi-CU decoder kernels are switched off (PNMN_DECODER_CLUSTER=0; the same
    branch serves a device too small for them) so the iteration falls back to decode(), and a pass that was prepared but
    not paired hands its drawn seed on: same sampled programs as the prepared pass would have produced."""
    from probnmn.data.synthetic import synthetic_batch
    from probnmn.models import ProgramGenerator
    from probnmn.vocabulary import Vocabulary

    dev = torch.device("cuda:0")
    vocab = Vocabulary.clevr()
    torch.manual_seed(6)
    pg = ProgramGenerator(vocab).to(dev)
    pg.train()
    q = synthetic_batch(vocab, 32, seed=43, with_image=False)["question"].to(dev)
    state = pg.encode(q)
    torch.manual_seed(9)
    prep = pg.decode_prepare(state, None, "sampling")
    assert prep is not None
    handed_on = pg.decode(state, None, "sampling", seed=prep["meta"]["seed"])["predictions"]
    torch.manual_seed(9)
    drawn = pg.decode(state, None, "sampling")["predictions"]
    assert torch.equal(handed_on, drawn)
    monkeypatch.setenv("PNMN_DECODER_CLUSTER", "0")
    assert pg.decode_prepare(state, None, "sampling") is None


def test_other_widths_run_step_by_step_and_say_so():
    """Shapes the persistent kernels are not built for (here hidden size 128; no reference config has one) take the
    step-by-step paths -- a GEMM and a cell launch per time step: same results as the oracle, and a RuntimeWarning the
    first time instead of a silent 10x slowdown."""
    import warnings

    from oracle import seq2seq_oracle as so
    from probnmn.models import ProgramGenerator
    from probnmn.modules import seq2seq_base
    from probnmn.vocabulary import Vocabulary

    vocab = Vocabulary.clevr()
    torch.manual_seed(3)
    model = ProgramGenerator(vocab, input_size=64, hidden_size=128, num_layers=2)
    vq, vp = vocab.get_vocab_size("questions"), vocab.get_vocab_size("programs")
    src, tgt = _tokens(6, 14, vq, 5), _tokens(6, 9, vp, 6)
    cpu_sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(DEV).train()
    ref_sd = {k: v.clone().requires_grad_(True) for k, v in cpu_sd.items()}
    ref = so.seq2seq_forward(ref_sd, src, tgt, "greedy")
    ref["loss"].mean().backward()
    seq2seq_base._SLOW_PATHS_NOTED.clear()
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        out = model(src.to(DEV), tgt.to(DEV), decoding_strategy="greedy")
        out["loss"].mean().backward()
        model(src.to(DEV), tgt.to(DEV), decoding_strategy="greedy")  # (said once per path, not once per call)
    said = [str(w.message) for w in caught if issubclass(w.category, RuntimeWarning) and "step by step" in str(w.message)]
    assert len(said) == 2 and any("LSTM layer" in m for m in said) and any("decoder" in m for m in said), said
    torch.testing.assert_close(out["loss"].detach().cpu(), ref["loss"].detach(), rtol=1e-4, atol=1e-4)
    assert torch.equal(out["predictions"].cpu(), ref["predictions"])
    _cmp_grads(model, ref_sd, "hidden128/tf")
