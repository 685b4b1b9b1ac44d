"""REINFORCE / ELBO arithmetic (CPU): the oracle and the product's ``probnmn.modules.elbo`` against
what the REAL reference produced (tests/golden/elbo_known.json)."""
import json
import os

import pytest
import torch

from oracle import elbo_oracle


@pytest.fixture(scope="module")
def known(golden_dir):
    with open(os.path.join(golden_dir, "elbo_known.json")) as f:
        return json.load(f)


def _leaves(known):
    return {k: torch.tensor(v, requires_grad=True) for k, v in known["fixed_losses"].items()}


def _duck(leaves, key, n):
    return lambda *a, **kw: {"predictions": torch.zeros(n, 3, dtype=torch.long), "loss": leaves[key]}


def test_hand_derived_known_answers(known):
    # logq=[-1,-2], rec=[-3,-1], R=[.5,1.5], beta=.1, decay=.99, b0=0
    first, second = known["calls"]
    assert first["elbo"] == pytest.approx(-0.4) and first["kl_divergence"] == pytest.approx(-1.6)
    assert first["baseline_after"] == pytest.approx(0.99)
    assert first["dneg_elbo_dlogq"] == pytest.approx([0.2, 0.7])
    assert second["kl_divergence"] == pytest.approx(-0.115, abs=1e-6) and second["baseline_after"] == pytest.approx(0.9999)


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_two_calls_match_reference(known, impl):
    from probnmn.modules import elbo as product

    if impl == "oracle":
        r = elbo_oracle.Reinforce(0.99)
        run = lambda lq, rec, rw: elbo_oracle.elbo_forward(r, 0.1, lq, rec, rw)  # noqa: E731
        base = lambda: r.baseline  # noqa: E731
    else:
        e = product._ElboWithReinforce(beta=0.1, baseline_decay=0.99)
        run = e._forward
        base = lambda: e._reinforce._reinforce_baseline  # noqa: E731
    for call in known["calls"]:
        logq = torch.tensor([-1.0, -2.0], requires_grad=True)
        rec = torch.tensor([-3.0, -1.0], requires_grad=True)
        out = run(logq, rec, torch.tensor([0.5, 1.5]))
        (-out["elbo"]).backward()
        for k in ("reconstruction_likelihood", "kl_divergence", "elbo", "reinforce_reward"):
            assert float(out[k]) == pytest.approx(call[k], abs=1e-6), k
        assert base() == pytest.approx(call["baseline_after"], abs=1e-6)
        assert logq.grad.tolist() == pytest.approx(call["dneg_elbo_dlogq"], abs=1e-6)
        assert rec.grad.tolist() == pytest.approx(call["dneg_elbo_drec"], abs=1e-6)


@pytest.mark.parametrize("objective", ["ours", "baseline"])
@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_joint_training_elbo_matches_reference(known, objective, impl):
    from probnmn.modules import elbo as product

    gold = known["joint_" + objective]
    lv = _leaves(known)
    n = len(known["fixed_losses"]["pg_loss"])
    if impl == "oracle":
        r = elbo_oracle.Reinforce(0.99)
        out = elbo_oracle.joint_training_elbo(r, 0.1, 1.0, objective, lv["pg_loss"], lv["qr_loss"], lv["prior_loss"], lv["nmn_loss"])
        baseline = r.baseline
    else:
        je = product.JointTrainingElbo(_duck(lv, "pg_loss", n), _duck(lv, "qr_loss", n), _duck(lv, "prior_loss", n),
                                       _duck(lv, "nmn_loss", n), beta=0.1, gamma=1.0, baseline_decay=0.99, objective=objective)
        out = je(None, None, None)
        baseline = je._reinforce._reinforce_baseline
    nmn_loss = out.pop("nmn_loss")
    (1.0 * nmn_loss - out["elbo"]).backward()
    assert float(nmn_loss) == pytest.approx(gold["nmn_loss"], abs=1e-6)
    for k, v in out.items():
        assert float(v) == pytest.approx(gold[k], abs=1e-5), k
    assert baseline == pytest.approx(gold["baseline_after"], abs=1e-5)
    for k, g in gold["grads"].items():
        if g is None or (impl == "product" and k == "prior_loss"):
            continue  # the product runs the frozen prior under no_grad (its gradient is unused: reward is detached)
        assert lv[k].grad is not None, k
        assert lv[k].grad.tolist() == pytest.approx(g, abs=1e-6), k


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_question_coding_elbo_matches_reference(known, impl):
    from probnmn.modules import elbo as product

    gold = known["question_coding"]
    lv = _leaves(known)
    n = len(known["fixed_losses"]["pg_loss"])
    if impl == "oracle":
        r = elbo_oracle.Reinforce(0.99)
        out = elbo_oracle.question_coding_elbo(r, 0.1, lv["pg_loss"], lv["qr_loss"], lv["prior_loss"])
        baseline = r.baseline
    else:
        qe = product.QuestionCodingElbo(_duck(lv, "pg_loss", n), _duck(lv, "qr_loss", n), _duck(lv, "prior_loss", n),
                                        beta=0.1, baseline_decay=0.99)
        out = qe(None)
        baseline = qe._reinforce._reinforce_baseline
    (-out["elbo"]).backward()
    for k, v in out.items():
        assert float(v) == pytest.approx(gold[k], abs=1e-5), k
    assert baseline == pytest.approx(gold["baseline_after"], abs=1e-5)
    for k in ("pg_loss", "qr_loss"):
        assert lv[k].grad.tolist() == pytest.approx(gold["grads"][k], abs=1e-6), k
