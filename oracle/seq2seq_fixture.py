"""Inputs, weights and gradient-digest positions of tests/golden/seq2seq_overrides.npz (test infrastructure): shared
by the generator (oracle/make_seq2seq_overrides.py, build container) and the tests that read the file on either box.
Everything is a function of the recorded seeds (oracle/detgen.py: numpy Philox), so the file stores outputs only."""
import zlib

import numpy as np
import torch

from oracle import detgen, seq2seq_oracle as so

SEEDS = {"pg": 4101, "qr": 4102, "prior": 4103, "inputs": 4110, "draws": 4120}
B, QLEN, PLEN = 8, 12, 9
V_Q, V_P = 100, 44  # 4 specials + 96 question words / 40 CLEVR program tokens (Vocabulary.clevr(), make_golden.namespaces())
GAIN = 2.0  # matrices at twice the kaiming scale: logits spread over ~+-1 instead of +-0.1 (arg-max margins >> round-off)
STEPS = {"pg": 26, "qr": 45}  # max_decoding_steps (reference models/program_generator.py:37, question_reconstructor.py:35)


def inputs():
    """Right-padded questions / programs with ragged lengths, incl. a one-token question and an EMPTY program."""
    gen = detgen.rng(SEEDS["inputs"])
    qlens = [12, 1, 7, 3, 10, 5, 2, 9]
    plens = [9, 0, 4, 1, 6, 8, 2, 5]
    q = np.zeros((B, QLEN), np.int64)
    p = np.zeros((B, PLEN), np.int64)
    for i in range(B):
        q[i, : qlens[i]] = gen.integers(4, V_Q, qlens[i])
        p[i, : plens[i]] = gen.integers(4, V_P, plens[i])
    return torch.from_numpy(q), torch.from_numpy(p)


def weights(kind: str) -> dict:
    """state_dict with the reference's parameter names for "pg" / "qr" / "prior"."""
    if kind == "prior":
        sd = detgen.fill_state_dict(so.prior_param_shapes(V_P), SEEDS[kind])
    else:
        v_src, v_tgt = (V_Q, V_P) if kind == "pg" else (V_P, V_Q)
        sd = detgen.fill_state_dict(so.seq2seq_param_shapes(v_src, v_tgt), SEEDS[kind])
    return {k: (v * GAIN if v.dim() >= 2 else v) for k, v in sd.items()}


def grad_entries(name: str, numel: int) -> np.ndarray:
    """The 128 positions of a parameter's flattened gradient the file stores."""
    return np.random.Generator(np.random.Philox(zlib.crc32(name.encode()))).integers(0, numel, 128)


def check_grads(g, prefix: str, named_grads, rtol: float) -> None:
    """Every gradient digest stored under ``prefix`` (l2 norm, sum, 128 entries) against ``named_grads``."""
    seen = 0
    for name, grad in named_grads:
        flat = grad.detach().reshape(-1).double().cpu().numpy()
        want_norm = float(g["%s/grad/%s/norm" % (prefix, name)])
        want_at = g["%s/grad/%s/at" % (prefix, name)].astype(np.float64)
        got_at = flat[grad_entries(name, flat.size)]
        scale = max(float(np.abs(want_at).max()), want_norm / np.sqrt(flat.size), 1e-12)
        assert np.abs(got_at - want_at).max() <= rtol * scale, (prefix, name, float(np.abs(got_at - want_at).max()), scale)
        assert abs(np.sqrt((flat * flat).sum()) - want_norm) <= rtol * want_norm + 1e-12, (prefix, name)
        assert abs(flat.sum() - float(g["%s/grad/%s/sum" % (prefix, name)])) <= rtol * (np.abs(flat).sum() + 1e-12), (prefix, name)
        seen += 1
    assert seen == sum(1 for k in g.files if k.startswith(prefix + "/grad/") and k.endswith("/norm")), prefix
