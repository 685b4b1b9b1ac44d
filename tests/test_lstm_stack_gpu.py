"""pnmn_lstm_stack_fwd / _bwd (csrc/lstm_stack.hip): the two layers of an encoder as a wavefront in one launch, several
encoders per launch -- against the per-layer entry points with the input projection / its data gradient as torch products.
A FIRST / TOP job must be bit-identical to pnmn_lstm_seq_fwd / _bwd; a SECOND / BELOW job agrees to fp32 round-off (2e-5 of
the largest entry; the products are associated differently)."""
import numpy as np
import pytest
import torch

from probnmn import _hip
from probnmn.modules.seq2seq_base import pack_fragments

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _layer(seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    r = lambda *s: (torch.rand(*s, device=DEV, generator=g) - 0.5) * 0.125  # noqa: E731
    return dict(w_ih=r(1024, 256), w_hh=r(1024, 256), b=r(1024))


def _seq_fwd(xp, tokens, w_hh, B, T):
    hs, cs, act = torch.empty(B, T, 256, device=DEV), torch.empty(B, T, 256, device=DEV), torch.empty(B, T, 1024, device=DEV)
    ws = torch.empty(int(_hip.lib().pnmn_lstm_seq_workspace_bytes(B, 0)), dtype=torch.uint8, device=DEV)
    wp = pack_fragments(w_hh)
    _hip.check(_hip.lib().pnmn_lstm_seq_fwd(xp.data_ptr(), tokens.data_ptr() if tokens is not None else None,
                                            tokens.stride(0) if tokens is not None else 0, wp.data_ptr(), hs.data_ptr(), cs.data_ptr(),
                                            act.data_ptr(), B, T, 256, ws.data_ptr(), _hip.stream_ptr(torch.device(DEV))), "lstm_seq_fwd")
    return hs, cs, act


def _seq_bwd(dhs, act, cs, w_hh, B, T):
    dg = torch.empty(B, T, 1024, device=DEV)
    ws = torch.empty(int(_hip.lib().pnmn_lstm_seq_workspace_bytes(B, 1)), dtype=torch.uint8, device=DEV)
    wt = pack_fragments(w_hh.t())
    _hip.check(_hip.lib().pnmn_lstm_seq_bwd(dhs.data_ptr(), act.data_ptr(), cs.data_ptr(), wt.data_ptr(), dg.data_ptr(), B, T, 256,
                                            ws.data_ptr(), _hip.stream_ptr(torch.device(DEV))), "lstm_seq_bwd")
    return dg


def _reference(enc, B, T, seed):
    """Separate launches: layer 1 from a token table, projection, layer 2; backward from a random output gradient."""
    g = torch.Generator(device=DEV).manual_seed(seed)
    V = 40
    table = (torch.rand(V, 1024, device=DEV, generator=g) - 0.5)
    tokens = torch.randint(0, V, (B, T), device=DEV, generator=g)
    l1, l2 = enc
    hs1, cs1, act1 = _seq_fwd(table, tokens, l1["w_hh"], B, T)
    xp2 = (hs1.double() @ l2["w_ih"].double().t() + l2["b"].double()).float().contiguous()
    hs2, cs2, act2 = _seq_fwd(xp2, None, l2["w_hh"], B, T)
    dhs2 = torch.randn(B, T, 256, device=DEV, generator=g)
    dg2 = _seq_bwd(dhs2, act2, cs2, l2["w_hh"], B, T)
    dhs1 = (dg2.double() @ l2["w_ih"].double()).float().contiguous()
    dg1 = _seq_bwd(dhs1, act1, cs1, l1["w_hh"], B, T)
    return dict(table=table, tokens=tokens, hs1=hs1, cs1=cs1, act1=act1, hs2=hs2, cs2=cs2, act2=act2, dhs2=dhs2, dg2=dg2, dg1=dg1)


def _stack(encs, refs, shapes):
    """All encoders' layers in one forward and one backward launch; returns per encoder the stack's tensors."""
    lib, st = _hip.lib(), _hip.stream_ptr(torch.device(DEV))
    n = 2 * len(encs)
    fj, bj = np.zeros(n, _hip.LSTM_STACK_JOB), np.zeros(n, _hip.LSTM_STACK_JOB)
    outs, keep = [], []
    for k, ((l1, l2), ref, (B, T)) in enumerate(zip(encs, refs, shapes)):
        o = {name: torch.full((B, T, w), float("nan"), device=DEV) for name, w in (("hs1", 256), ("cs1", 256), ("act1", 1024), ("hs2", 256),
                                                                                  ("cs2", 256), ("act2", 1024), ("dg1", 1024), ("dg2", 1024))}
        packs = [pack_fragments(l1["w_hh"]), pack_fragments(l2["w_hh"]), pack_fragments(l2["w_ih"]), pack_fragments(l1["w_hh"].t()),
                 pack_fragments(l2["w_hh"].t()), pack_fragments(l2["w_ih"].t())]
        keep.append(packs)
        a, b = fj[2 * k], fj[2 * k + 1]
        a["xp"], a["tokens"], a["token_stride"], a["w_hh"] = ref["table"].data_ptr(), ref["tokens"].data_ptr(), T, packs[0].data_ptr()
        a["hs"], a["cs"], a["act"], a["B"], a["T"], a["dep"] = o["hs1"].data_ptr(), o["cs1"].data_ptr(), o["act1"].data_ptr(), B, T, -1
        b["w_hh"], b["w_ih"], b["bias"] = packs[1].data_ptr(), packs[2].data_ptr(), l2["b"].data_ptr()
        b["hs"], b["cs"], b["act"], b["B"], b["T"], b["dep"] = o["hs2"].data_ptr(), o["cs2"].data_ptr(), o["act2"].data_ptr(), B, T, 2 * k
        # backward: job 2k = layer 2 (top), job 2k + 1 = layer 1 (below it)
        a, b = bj[2 * k], bj[2 * k + 1]
        a["dhs"], a["act"], a["cs"], a["w_hh"], a["dgates"] = ref["dhs2"].data_ptr(), ref["act2"].data_ptr(), ref["cs2"].data_ptr(), packs[4].data_ptr(), o["dg2"].data_ptr()
        a["B"], a["T"], a["dep"] = B, T, -1
        b["act"], b["cs"], b["w_hh"], b["w_ih"], b["dgates"] = ref["act1"].data_ptr(), ref["cs1"].data_ptr(), packs[3].data_ptr(), packs[5].data_ptr(), o["dg1"].data_ptr()
        b["B"], b["T"], b["dep"] = B, T, 2 * k
        outs.append(o)
    for jobs, fn, backward in ((fj, lib.pnmn_lstm_stack_fwd, 0), (bj, lib.pnmn_lstm_stack_bwd, 1)):
        nbytes = int(lib.pnmn_lstm_stack_workspace_bytes(jobs.ctypes.data, n, backward))
        assert nbytes > 0
        ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
        for _ in range(2):  # (twice: the counters must come back to zero)
            _hip.check(fn(jobs.ctypes.data, n, ws.data_ptr(), st), "lstm_stack")
        torch.cuda.synchronize()
    return outs


def _close(got, want, tol=2e-5):
    assert torch.isfinite(got).all()
    return float((got - want).abs().max()) / (float(want.abs().max()) + 1e-30) < tol


@pytest.mark.parametrize("shapes", [[(128, 46)], [(20, 5)], [(128, 27), (64, 28)], [(256, 12)], [(33, 1), (50, 9)]])
def test_stack_equals_separate_layers(shapes):
    encs = [(_layer(10 * k + 1), _layer(10 * k + 2)) for k in range(len(shapes))]
    refs = [_reference(enc, B, T, 100 + k) for k, (enc, (B, T)) in enumerate(zip(encs, shapes))]
    outs = _stack(encs, refs, shapes)
    for o, ref in zip(outs, refs):
        for name in ("hs1", "cs1", "act1", "dg2"):  # FIRST / TOP jobs: the per-layer kernels' arithmetic
            assert torch.equal(o[name], ref[name]), name
        for name in ("hs2", "cs2", "act2", "dg1"):
            assert _close(o[name], ref[name]), name


def test_does_not_fit_reports_zero_workspace():
    jobs = np.zeros(2, _hip.LSTM_STACK_JOB)
    jobs["B"], jobs["T"], jobs["dep"] = 1024, 10, -1
    assert int(_hip.lib().pnmn_lstm_stack_workspace_bytes(jobs.ctypes.data, 2, 0)) == 0


def test_rate_report(capsys):
    B, T = 128, 46
    enc = [(_layer(1), _layer(2))]
    ref = [_reference(enc[0], B, T, 5)]
    import time
    lib, st = _hip.lib(), _hip.stream_ptr(torch.device(DEV))

    def clock(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    t_sep = clock(lambda: _reference(enc[0], B, T, 5))
    t_stack = clock(lambda: _stack(enc, ref, [(B, T)]))
    with capsys.disabled():
        print("encoder forward + backward, %d rows x %d steps: separate launches (+ torch fp64 projections) %.0f us, stack (x2 runs) %.0f us"
              % (B, T, t_sep, t_stack))
