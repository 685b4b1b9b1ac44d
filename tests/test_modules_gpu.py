"""The standalone module classes (probnmn.modules.nmn_modules) on the MI355X against the vectors
the REAL reference produced (tests/golden/nmn_modules_full.npz) and, for gradients, against the
oracle's autograd.  Tolerance 1e-4 absolute on O(1) activations (fp32 accumulation order)."""
import os

import numpy as np
import pytest
import torch

from fixtures import full_module_inputs

pytestmark = pytest.mark.gpu

SIZES = pytest.mark.parametrize("size", [14, 28], ids=["14x14", "28x28"])


def _build(cls, tok, sd, dev):
    m = cls(128)
    own = {k[len(tok) + 1:]: v for k, v in sd.items() if k.startswith(tok + ".")}
    m.load_state_dict(own)
    return m.to(dev)


@SIZES
def test_modules_match_reference_golden(golden_dir, size):
    from probnmn.modules import nmn_modules as M

    gold = np.load(os.path.join(golden_dir, "nmn_modules_full%s.npz" % ("" if size == 14 else "_%d" % size)))
    feats, feats2, attn, attn2, toks, sd = full_module_inputs(size)
    dev = torch.device("cuda:0")
    f, f2, a, a2 = (t.to(dev) for t in (feats, feats2, attn, attn2))
    with torch.no_grad():
        got = {
            "and_1_1": M.AndModule()(a, a2),
            "or_1_1": M.OrModule()(a, a2),
            "and_1_128": M.AndModule()(a, f),
            "or_128_128": M.OrModule()(f, f2),
            "attention": _build(M.AttentionModule, toks["attention"], sd, dev)(f, a),
            "query": _build(M.QueryModule, toks["query"], sd, dev)(f, a),
            "relate": _build(M.RelateModule, toks["relate"], sd, dev)(f, a),
            "same": _build(M.SameModule, toks["same"], sd, dev)(f, a),
            "comparison": _build(M.ComparisonModule, toks["comparison"], sd, dev)(f, f2),
        }
    for k, v in got.items():
        assert tuple(v.shape) == gold[k].shape, k
        if k.startswith(("and", "or")):
            assert np.array_equal(v.cpu().numpy(), gold[k]), k  # min/max are exact
        else:
            np.testing.assert_allclose(v.cpu().numpy(), gold[k], rtol=1e-4, atol=1e-4, err_msg=k)


@SIZES
@pytest.mark.parametrize("kind", ["attention", "query", "relate", "same", "comparison", "and"])
def test_module_gradients_match_oracle(kind, size):
    from oracle import nmn_oracle
    from probnmn.modules import nmn_modules as M

    feats, feats2, attn, attn2, toks, sd = full_module_inputs(size)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)

    def leaf(t):
        return t.clone().requires_grad_(True), t.clone().to(dev).requires_grad_(True)

    f_c, f_g = leaf(feats)
    x_c, x_g = leaf(feats2 if kind in ("comparison",) else attn)
    if kind == "and":
        mod = M.AndModule()
        out_g = mod(x_g, f_g)
        out_c = nmn_oracle.and_module(x_c, f_c)
        params_c = {}
    else:
        tok = toks[kind]
        cls = {"attention": M.AttentionModule, "query": M.QueryModule, "relate": M.RelateModule,
               "same": M.SameModule, "comparison": M.ComparisonModule}[kind]
        mod = _build(cls, tok, sd, dev)
        params_c = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith(tok + ".")}
        fn = {"attention": nmn_oracle.attention_module, "query": nmn_oracle.query_module,
              "relate": nmn_oracle.relate_module, "same": nmn_oracle.same_module,
              "comparison": nmn_oracle.comparison_module}[kind]
        out_c = fn(params_c, tok, f_c, x_c)
        out_g = mod(f_g, x_g)
    dout = torch.randn(out_c.shape, generator=g)
    out_c.backward(dout)
    out_g.backward(dout.to(dev))

    def close(got, want, name):
        scale = float(want.abs().max()) + 1e-12
        err = float((got.cpu() - want).abs().max()) / scale
        assert err < 2e-3, (kind, name, err)

    close(f_g.grad, f_c.grad, "d feats")
    close(x_g.grad, x_c.grad, "d second input")
    for k, v in params_c.items():
        p = dict(mod.named_parameters())[k.split(".", 1)[1]]
        close(p.grad, v.grad, k)


def test_unsupported_shapes_fail_loudly():
    from probnmn import _hip
    from probnmn.modules import nmn_modules as M

    dev = torch.device("cuda:0")
    with pytest.raises(NotImplementedError):
        M.AttentionModule(64).to(dev)(torch.zeros(1, 64, 14, 14, device=dev), torch.ones(1, 1, 14, 14, device=dev))
    with pytest.raises(NotImplementedError):
        M.AttentionModule(128).to(dev)(torch.zeros(1, 128, 20, 20, device=dev), torch.ones(1, 1, 20, 20, device=dev))
    with pytest.raises(_hip.HipLibraryError):
        M.AttentionModule(128)(torch.zeros(1, 128, 14, 14), torch.ones(1, 1, 14, 14))
