"""NeuralModuleNetwork on the MI355X against the CPU oracle at the reference's full dimensions
(1024x14x14 features, 128 module channels): same weights, same programs, same answers.
Tolerance: fp32 accumulation order only (relative 1e-3 on gradients that sum ~1e5 products)."""
import numpy as np
import pytest
import torch

from fixtures import LONG_CASES, VALIDITY_CASES, encode_programs

pytestmark = pytest.mark.gpu


def _setup(seed=0, cases=VALIDITY_CASES, size=14, length=26):
    from probnmn.models.nmn import NeuralModuleNetwork
    from probnmn.vocabulary import Vocabulary

    vocab = Vocabulary.clevr()
    torch.manual_seed(seed)
    net = NeuralModuleNetwork(vocab, image_feature_size=(1024, size, size))
    programs = encode_programs(cases, vocab.get_token_to_index_vocabulary("programs"), length=length)
    B = programs.size(0)
    g = torch.Generator().manual_seed(seed + 1)
    features = torch.relu(torch.randn(B, 1024, size, size, generator=g))
    answers = torch.randint(0, 28, (B,), generator=g)
    return vocab, net, programs, features, answers


def _oracle(vocab, sd, programs, features, answers):
    from oracle import nmn_oracle

    sd = {k: v.detach().clone().contiguous().requires_grad_(True) for k, v in sd.items()}
    out = nmn_oracle.nmn_forward(sd, vocab.get_index_to_token_vocabulary("programs"), features, programs, answers)
    if answers is not None:
        out["loss"].mean().backward()
    return out, sd


def _grad_errors(net, ref_sd, l2=None):
    """max |g_hip - g_oracle| / max |g_oracle| per parameter (``l2``: dict that receives the relative
    l2 error per parameter)."""
    errs = {}
    for name, p in net.named_parameters():
        g_ref = ref_sd[name].grad
        g_ref = torch.zeros_like(ref_sd[name]) if g_ref is None else g_ref
        got = torch.zeros_like(g_ref) if p.grad is None else p.grad.detach().cpu()
        scale = float(g_ref.abs().max())
        if scale == 0.0:
            assert float(got.abs().max()) == 0.0, name  # unused module: exactly no gradient
            continue
        errs[name] = float((got - g_ref).abs().max()) / scale
        if l2 is not None:
            l2[name] = float((got - g_ref).norm() / g_ref.norm())
    return errs


def test_forward_backward_matches_oracle_full_batch():
    """All 36 golden programs (valid and invalid) in one batch: identical predictions, losses to
    1e-4, gradients to 2e-2 of each tensor's max (see the per-program test for why not tighter)."""
    vocab, net, programs, features, answers = _setup()
    cpu_sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    ref, ref_sd = _oracle(vocab, cpu_sd, programs, features, answers)

    dev = torch.device("cuda:0")
    net.to(dev).train()
    out = net(features.to(dev), programs.to(dev), answers.to(dev))
    out["loss"].mean().backward()
    torch.cuda.synchronize()

    assert torch.equal(out["predictions"].cpu(), ref["predictions"])
    torch.testing.assert_close(out["loss"].detach().cpu(), ref["loss"].detach(), rtol=1e-4, atol=1e-4)
    n_invalid = int((ref["valid"] == 0).sum())
    assert n_invalid > 0
    assert out["metrics"]["average_invalid"] == n_invalid
    errs = _grad_errors(net, ref_sd)
    assert len(errs) > 60
    worst = max(errs, key=errs.get)
    print("worst relative gradient error", worst, errs[worst])
    assert errs[worst] < 2e-2, (worst, errs[worst])

    # state_dict keys and logical shapes are the reference's
    for k, v in net.state_dict().items():
        assert tuple(v.shape) == tuple(cpu_sd[k].shape), k
        torch.testing.assert_close(v.cpu(), cpu_sd[k])


def test_gradients_per_program_tight():
    """Each valid golden program on its own (batch of 2, fresh features).

    The network is piecewise linear with hard gates (ReLU, 2x2 max-pool arg-max, min/max): when a
    pre-activation lands within fp32 round-off of a gate, the GPU (MFMA accumulation order) and the
    CPU oracle can take different sides, and that one element's gradient is routed differently --
    a property of fp32, not of the kernels (the kernel-level tests in test_hip_kernels.py, which
    share saved activations with autograd, match to 1e-4 everywhere).  A program runs ~4e5 gated
    elements whose pre-activations are sums of ~1 152 products (round-off ~1e-6 on O(1) values), so
    about one program in three holds a flipped gate, WHICH ones depends on the summation order
    (scripts/r04_dbg3.py: the round-3 and round-4 convolution kernels, same loss bit for bit, differ from
    each other by 1e-2 .. 5e-2 on 7 of 21 programs and by ~1e-6 on the rest), and in a batch of two one
    flipped element moves a tensor's gradient by up to ~5e-2 of its maximum.  So a program whose input
    misses the tight bar gets further independent inputs, up to eight: an arithmetic or indexing error
    shows on every input, a flipped gate on one in three.  The bar: EVERY program matches on EVERY
    parameter to 2e-4 of the tensor's max on at least one input, the typical program to 5e-5 on its
    first, and at least half of the programs to 2e-4 on their first."""
    from probnmn.models.nmn import NeuralModuleNetwork  # noqa: F401

    vocab, net, _, _, _ = _setup()
    cpu_sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    stoi = vocab.get_token_to_index_vocabulary("programs")
    dev = torch.device("cuda:0")
    net.to(dev).train()
    g = torch.Generator().manual_seed(1)
    first, best, all_tries = [], [], []
    for case in VALIDITY_CASES:
        programs = encode_programs([case, case], stoi)
        tries = []
        for _ in range(8):
            features = torch.relu(torch.randn(2, 1024, 14, 14, generator=g))
            answers = torch.randint(0, 28, (2,), generator=g)
            ref, ref_sd = _oracle(vocab, cpu_sd, programs, features, answers)
            if int(ref["valid"][0]) == 0:
                break
            net.zero_grad(set_to_none=True)
            out = net(features.to(dev), programs.to(dev), answers.to(dev))
            out["loss"].mean().backward()
            torch.testing.assert_close(out["loss"].detach().cpu(), ref["loss"].detach(), rtol=1e-5, atol=2e-6)
            tries.append(max(_grad_errors(net, ref_sd).values()))
            if tries[-1] < 2e-4:
                break
        if tries:
            first.append(tries[0])
            best.append(min(tries))
            all_tries.extend(tries)
    w = np.sort(np.asarray(first))
    print("per-program worst gradient errors (first input):", np.array2string(w, precision=1))
    print("                                  (best of <= 8):", np.array2string(np.sort(np.asarray(best)), precision=1))
    assert len(w) >= 20
    assert np.median(w) < 5e-5
    assert max(best) < 2e-4
    # The flip RATE is bounded too ("best of eight" alone would pass a kernel that rounded differently on every input):
    # measured one input in three above the tight bar (rounds 3-5, two different convolution kernels); the bar is that
    # rate with a margin, over the first inputs and over every input tried.
    flip_first = float(np.mean(w >= 2e-4))
    flip_all = float(np.mean(np.asarray(all_tries) >= 2e-4))
    print("flip rate: %.2f of first inputs, %.2f of all %d inputs" % (flip_first, flip_all, len(all_tries)))
    assert flip_first <= 0.45, flip_first
    assert flip_all <= 0.5, flip_all


# BASELINE config 5: 28x28 feature maps, programs of up to 40 tokens.  A slice of the golden cases that
# uses every module kind, both verdicts and every long chain (the oracle's classifier alone is
# 200704 x 1024 weights: the CPU side of this test is what bounds the batch).
CONFIG5_CASES = [VALIDITY_CASES[i] for i in (0, 3, 9, 11, 12, 13, 14, 15, 16, 17, 22, 34)] + LONG_CASES


def test_config5_28x28_long_programs_match_oracle():
    vocab, net, programs, features, answers = _setup(seed=5, cases=CONFIG5_CASES, size=28, length=40)
    cpu_sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    assert tuple(cpu_sd["classifier.4.weight"].shape) == (1024, 1024 * 14 * 14)
    ref, ref_sd = _oracle(vocab, cpu_sd, programs, features, answers)

    dev = torch.device("cuda:0")
    net.to(dev).train()
    out = net(features.to(dev), programs.to(dev), answers.to(dev))
    out["loss"].mean().backward()
    torch.cuda.synchronize()
    assert torch.equal(out["predictions"].cpu(), ref["predictions"])
    torch.testing.assert_close(out["loss"].detach().cpu(), ref["loss"].detach(), rtol=1e-4, atol=1e-4)
    n_invalid = int((ref["valid"] == 0).sum())
    assert 0 < n_invalid < len(CONFIG5_CASES)
    assert out["metrics"]["average_invalid"] == n_invalid
    l2 = {}
    errs = _grad_errors(net, ref_sd, l2)
    assert len(errs) > 60
    worst = max(errs, key=errs.get)
    v = np.asarray(list(errs.values()))
    print("28x28 gradient errors: median max-rel %.1e, worst max-rel %.1e (%s), worst l2-rel %.1e"
          % (np.median(v), errs[worst], worst, max(l2.values())))
    # Hard gates (ReLU, max-pool arg-max, min/max) within round-off of a tie route one element's gradient
    # differently on the two sides; at this size one example in four has a unit of the 200704-input hidden
    # layer inside the round-off of its 200k-term dot product, which perturbs that example's whole
    # gradient by ~4e-3 in l2 (see tests/test_nmn_per_module_gpu.py, which holds the tight per-module bars).
    # Flip-proof bar here: every tensor within 5e-2 in relative l2 (an indexing error gives >= 0.14).
    assert max(l2.values()) < 5e-2, max(l2, key=l2.get)
    assert np.median(v) < 1e-3
    assert errs[worst] < 1e-1, (worst, errs[worst])
    # evaluation pass (no answers, no gradient) on the same network
    net.eval()
    with torch.no_grad():
        ev = net(features.to(dev), programs.to(dev))
    ref_ev, _ = _oracle(vocab, cpu_sd, programs, features, None)
    assert torch.equal(ev["predictions"].cpu(), ref_ev["predictions"])
    torch.testing.assert_close(ev["loss"].cpu(), ref_ev["loss"].detach(), rtol=1e-4, atol=1e-4)


def test_launch_trace_reports_the_shipped_path():
    """engine.begin_trace / end_trace (pnmn_launch_trace_*): the library times the conv / weight-gradient launches of its own
    lists -- stem, planner forward, backward -- and accounts their algorithmic work; the step's results do not change."""
    vocab, net, programs, features, answers = _setup(seed=5, cases=VALIDITY_CASES[:16])
    dev = torch.device("cuda:0")
    net.to(dev).train()
    B = features.size(0)

    def step():
        net.zero_grad(set_to_none=True)
        out = net(features.to(dev), programs.to(dev), answers.to(dev))
        out["loss"].mean().backward()
        torch.cuda.synchronize()
        return out["loss"].detach().clone()

    loss0 = step()
    net.engine.begin_trace()
    loss1 = step()
    events = net.engine.end_trace()
    assert torch.equal(loss0, loss1)
    assert net.engine.end_trace() == []  # (collected: the trace is off and empty)
    sites = [e[1] for e in events]
    assert sites[:2] == ["stem conv1", "stem conv2"] and sites[-3:] == ["stem conv2 wgrad", "stem conv2 dgrad", "stem conv1 wgrad"]
    assert sites.count("classifier conv") == 1 and sites.count("classifier dgrad") == 1 and sites.count("classifier wgrad") == 1
    n_conv, n_proj = net.engine.last_counts
    assert n_conv > 0 and ("module conv" in sites) and ("module dgrad" in sites) and ("module wgrad" in sites)
    by = {w: [e for e in events if e[1] == w] for w in set(sites)}
    hw, c = 14 * 14, 128
    cin = features.size(1)
    assert by["stem conv1"][0][2] == 2.0 * B * hw * c * 9 * cin and by["stem conv1 wgrad"][0][2] == 2.0 * B * hw * c * 9 * cin
    assert by["stem conv2"][0][2] == 2.0 * B * hw * c * 9 * c
    # algorithmic bytes: every input chunk and the output once per item, each distinct weight once per call (the items are
    # the ones the launch saw: copied in stream order, not read back after the step has re-used their buffer)
    assert by["stem conv1"][0][4] == B * (cin // c + 1) * hw * c * 4.0 + c * 9 * cin * 4.0
    assert by["stem conv2"][0][4] == B * 2 * hw * c * 4.0 + c * 9 * c * 4.0
    # every module conv once forward, once as a data gradient (or not at all where its input needs none), once in the
    # deferred weight gradient; dilation-8 items count 78 / 117 of the taps, so compare with the record count only from above
    full = 2.0 * hw * c * 9 * c
    fwd = sum(e[2] for e in by["module conv"])
    assert 0.6 * n_conv * full <= fwd <= n_conv * full
    assert sum(e[2] for e in by["module wgrad"]) == n_conv * full
    assert all(e[3] > 0 and e[4] > 0 and e[0] in ("conv_nhwc", "conv_wgrad") for e in events)


def test_eval_without_answers_and_repeat_is_deterministic():
    vocab, net, programs, features, _ = _setup(seed=3, cases=VALIDITY_CASES[:12])
    cpu_sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    ref, _ = _oracle(vocab, cpu_sd, programs, features, None)
    dev = torch.device("cuda:0")
    net.to(dev).eval()
    with torch.no_grad():
        out1 = net(features.to(dev), programs.to(dev))
        out2 = net(features.to(dev), programs.to(dev))
    assert "metrics" not in out1
    assert torch.equal(out1["predictions"].cpu(), ref["predictions"])
    torch.testing.assert_close(out1["loss"].cpu(), ref["loss"].detach(), rtol=1e-4, atol=1e-4)
    assert torch.equal(out1["loss"], out2["loss"])  # forward has no atomics: bitwise repeatable


def test_cpu_tensors_fail_loudly():
    from probnmn import _hip

    vocab, net, programs, features, answers = _setup(cases=VALIDITY_CASES[:2])
    with pytest.raises(_hip.HipLibraryError):
        net(features, programs, answers)


@pytest.mark.parametrize("size", [14, 28])
def test_row_subset_equals_the_gathered_batch(size):
    """``forward(features, ..., rows=idx)`` (the layout kernel reads through the index) against
    ``forward(features[idx], ...)``: same kernels on the same values -- losses and predictions bit for bit, the stem's
    weight gradient (the only gradient that reads the input features) to summation order."""
    cases = VALIDITY_CASES if size == 14 else LONG_CASES
    vocab, net, programs, features, answers = _setup(seed=3, cases=cases, size=size, length=26 if size == 14 else 40)
    dev = torch.device("cuda:0")
    net.to(dev)
    B = programs.size(0)
    g = torch.Generator().manual_seed(11)
    big = torch.relu(torch.randn(B + 5, 1024, size, size, generator=g)).to(dev)
    idx = torch.randperm(B + 5, generator=g)[:B].to(dev)
    outs = []
    for use_rows in (False, True):
        net.zero_grad()
        net.train()
        if use_rows:
            out = net(big, programs, answers.to(dev), rows=idx)
        else:
            out = net(big[idx], programs, answers.to(dev))
        out["loss"].mean().backward()
        outs.append((out["loss"].detach().clone(), out["predictions"].clone(), net.stem[0].weight.grad.clone(),
                     net.classifier[6].weight.grad.clone()))
    (loss_a, pred_a, gs_a, gc_a), (loss_b, pred_b, gs_b, gc_b) = outs
    assert torch.equal(loss_a, loss_b) and torch.equal(pred_a, pred_b)
    for a, b in ((gs_a, gs_b), (gc_a, gc_b)):  # (weight gradients are summed with fp32 atomics: order varies run to run)
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())


def test_mask_backward_modes_agree():
    """The three ways the backward of `feats * attn` is scheduled (probnmn.runtime.schedule: 2 = d(attention) in the data
    gradient's epilogue + ONE deferred gather of d(feats), the default; 1 = both fused into the epilogue; 0 = a separate
    kernel per level) give the same gradients."""
    from probnmn.data.synthetic import synthetic_batch
    from probnmn.models.nmn import NeuralModuleNetwork
    from probnmn.vocabulary import Vocabulary

    dev = torch.device("cuda:0")
    vocab = Vocabulary.clevr()
    batch = synthetic_batch(vocab, 24, seed=21)
    images, answers = batch["image"].to(dev), batch["answer"].to(dev)
    results = {}
    for mode in (2, 1, 0):
        torch.manual_seed(7)
        net = NeuralModuleNetwork(vocab, class_projection_channels=128, classifier_linear_size=64).to(dev)
        net.engine.ensure_arena()
        net.engine.planner_config.fuse_mask_bwd = mode
        net.train()
        out = net(images, batch["program"], answers)
        out["loss"].mean().backward()
        torch.cuda.synchronize()
        results[mode] = (out["loss"].detach().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters()})
    loss0, grads0 = results[2]
    for key, (loss, grads) in results.items():
        assert torch.equal(loss, loss0), key
        for n, g in grads.items():
            scale = float(grads0[n].abs().max()) + 1e-12
            assert float((g - grads0[n]).abs().max()) <= 3e-5 * scale, (key, n)


@pytest.mark.gpu
@pytest.mark.parametrize("n, deep, cus", [(24, False, 0), (200, False, 0), (96, True, 192), (700, False, 0)])
def test_conv_splits_give_bit_identical_activations(n, deep, cus):
    """The streamed convolution (csrc/conv_stream.h) splits OUTPUT work only -- a workgroup computes 128 / 64 / 32 / 16 of
    a unit's channels, a wave one or two 16-channel tiles x 13 / 7 / 4 m-tiles -- and every wave contracts all input
    channels of its outputs in the same order, so the split the launch planner picks (by launch size and CU budget) must
    not change a single bit of the forward pass: the losses of the whole network are EQUAL under split 1, 2, 4, 6, 8, 14, 26 and
    under the planner's own choice, and the gradients differ by the order of the weight gradients' fp32 atomic adds
    alone (<= 1e-4 of a tensor's largest entry)."""
    from probnmn import _hip
    from probnmn.data.synthetic import synthetic_batch
    from probnmn.models.nmn import NeuralModuleNetwork
    from probnmn.vocabulary import Vocabulary

    dev = torch.device("cuda:0")
    vocab = Vocabulary.clevr()
    batch = synthetic_batch(vocab, n, seed=31 + n, deep=deep)
    images, answers = batch["image"].to(dev), batch["answer"].to(dev)
    results = {}
    try:
        for split in (1, 2, 4, 6, 8, 14, 26, 0):
            _hip.check(_hip.lib().pnmn_conv_force_split(split), "force split")
            torch.manual_seed(7)
            net = NeuralModuleNetwork(vocab, class_projection_channels=128, classifier_linear_size=64).to(dev)
            net.engine.ensure_arena()
            net.engine.conv_cus = cus
            net.train()
            net.zero_grad(set_to_none=True)
            out = net(images, batch["program"], answers)
            out["loss"].mean().backward()
            torch.cuda.synchronize()
            results[split] = (out["loss"].detach().clone(), out["predictions"].clone(),
                              {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})
    finally:
        _hip.lib().pnmn_conv_force_split(0)
    loss1, pred1, grads1 = results[1]
    for split, (loss, pred, grads) in results.items():
        assert torch.equal(pred, pred1), split
        assert torch.equal(loss, loss1), split
        assert grads.keys() == grads1.keys() and len(grads) > 20
        for k, g in grads.items():
            scale = float(grads1[k].abs().max()) + 1e-12
            assert float((g - grads1[k]).abs().max()) <= 1e-4 * scale, (split, k)




def test_first_fully_connected_layer_on_own_gemm_equals_the_library_path(monkeypatch):
    """models.nmn._first_fc with PNMN_FC_OWN_ROWS set (opt-in: pnmn_gemm for the 50 176 -> 1024 layer) against the default
    library path: output, d(input), weight and bias gradients at a row count with a ragged last tile."""
    import torch.nn as nn
    from probnmn.models import nmn as M

    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    layer = nn.Linear(50176, 1024).to(dev)
    x = torch.randn(200, 50176, device=dev)
    dy = torch.randn(200, 1024, device=dev)
    outs = []
    for rows in (0, 1):
        monkeypatch.setattr(M, "OWN_FC_ROWS", rows)
        layer.zero_grad()
        xi = x.clone().requires_grad_(True)
        y = M._first_fc(layer, xi)
        y.backward(dy)
        outs.append((y.detach(), xi.grad, layer.weight.grad.clone(), layer.bias.grad.clone()))
        with torch.no_grad():
            assert torch.allclose(M._first_fc(layer, x), y.detach(), rtol=0, atol=1e-5 * float(y.abs().max()))
    for got, want in zip(outs[1], outs[0]):
        assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())
