"""The CPU oracle against the vectors the real reference produced (oracle/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import nmn_oracle

from fixtures import full_module_inputs, small_network_inputs


SIZES = pytest.mark.parametrize("size", [14, 28], ids=["14x14", "28x28"])


def _suffix(size):
    return "" if size == 14 else "_%d" % size


@SIZES
def test_modules_match_reference_vectors(golden_dir, size):
    gold = np.load(os.path.join(golden_dir, "nmn_modules_full%s.npz" % _suffix(size)))
    feats, feats2, attn, attn2, toks, sd = full_module_inputs(size)
    with torch.no_grad():
        mine = {
            "and_1_1": nmn_oracle.and_module(attn, attn2),
            "or_1_1": nmn_oracle.or_module(attn, attn2),
            "and_1_128": nmn_oracle.and_module(attn, feats),
            "or_128_128": nmn_oracle.or_module(feats, feats2),
            "attention": nmn_oracle.attention_module(sd, toks["attention"], feats, attn),
            "query": nmn_oracle.query_module(sd, toks["query"], feats, attn),
            "relate": nmn_oracle.relate_module(sd, toks["relate"], feats, attn),
            "same": nmn_oracle.same_module(sd, toks["same"], feats, attn),
            "comparison": nmn_oracle.comparison_module(sd, toks["comparison"], feats, feats2),
        }
    for k, v in mine.items():
        # same torch build on both boxes, but CPU conv kernels may pick another blocking on
        # another core count: allow fp32 round-off, not more.
        np.testing.assert_allclose(v.numpy(), gold[k], rtol=1e-5, atol=1e-6, err_msg=k)


@SIZES
def test_network_matches_reference_vectors(golden_dir, size):
    """size 28 = BASELINE config 5: 28x28 maps and programs of up to 40 tokens."""
    gold = np.load(os.path.join(golden_dir, "nmn_small%s.npz" % _suffix(size)))
    ns, programs, features, answers, sd = small_network_inputs(size)
    assert programs.size(1) == (26 if size == 14 else 40)
    itos = dict(enumerate(ns["programs"]))
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = nmn_oracle.nmn_forward(sd, itos, features, programs, answers)
    out["loss"].mean().backward()
    assert np.array_equal(out["predictions"].numpy(), gold["predictions"])
    assert np.array_equal(out["valid"].numpy(), gold["valid"])
    np.testing.assert_allclose(out["loss"].detach().numpy(), gold["loss"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out["logits"].detach().numpy(), gold["logits"], rtol=1e-5, atol=1e-6)
    for k, p in sd.items():
        g = p.grad.numpy() if p.grad is not None else np.zeros(p.shape, np.float32)
        np.testing.assert_allclose(g, gold["grad::" + k], rtol=1e-4, atol=1e-6, err_msg=k)
    with torch.no_grad():
        noans = nmn_oracle.nmn_forward(sd, itos, features, programs, None)
    np.testing.assert_allclose(noans["loss"].numpy(), gold["loss_without_answers"], rtol=1e-5, atol=1e-6)
    # invalid programs: constant loss 3.33, prediction = @@UNKNOWN@@ (28)
    inv = gold["valid"] == 0
    assert inv.any()
    assert np.all(gold["loss"][inv] == np.float32(3.33)) and np.all(gold["predictions"][inv] == 28)
