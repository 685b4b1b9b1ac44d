"""The fused REINFORCE / ELBO kernel (pnmn_elbo_rows behind probnmn.modules.elbo.*.combine on device
tensors) against what the REAL reference produced (tests/golden/elbo_known.json): values, the moving
baseline over two calls, and the gradients of the trainer's objective."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def known(golden_dir):
    with open(os.path.join(golden_dir, "elbo_known.json")) as f:
        return json.load(f)


def _leaves(known):
    return {k: torch.tensor(v, device=DEV, requires_grad=True) for k, v in known["fixed_losses"].items()}


def test_joint_training_ours_fused(known):
    from probnmn.modules.elbo import JointTrainingElbo

    gold = known["joint_ours"]
    lv = _leaves(known)
    je = JointTrainingElbo(None, None, None, None, beta=0.1, gamma=1.0, baseline_decay=0.99, objective="ours")
    out = je.combine(lv["pg_loss"], lv["qr_loss"], lv["prior_loss"].detach(), {"loss": lv["nmn_loss"]})
    nmn_loss = out.pop("nmn_loss")
    (1.0 * nmn_loss - out["elbo"]).backward()
    assert float(nmn_loss) == pytest.approx(gold["nmn_loss"], abs=1e-6)
    for k, v in out.items():
        assert float(v) == pytest.approx(gold[k], abs=1e-5), k
    assert je._reinforce._reinforce_baseline == pytest.approx(gold["baseline_after"], abs=1e-5)
    for k in ("pg_loss", "qr_loss", "nmn_loss"):
        assert lv[k].grad.tolist() == pytest.approx(gold["grads"][k], abs=1e-6), k


def test_question_coding_fused_and_second_call_uses_the_moved_baseline(known):
    from oracle import elbo_oracle
    from probnmn.modules.elbo import QuestionCodingElbo

    gold = known["question_coding"]
    lv = _leaves(known)
    qe = QuestionCodingElbo(None, None, None, beta=0.1, baseline_decay=0.99)
    out = qe.combine(lv["pg_loss"], lv["qr_loss"], lv["prior_loss"].detach())
    (-out["elbo"]).backward()
    for k, v in out.items():
        assert float(v) == pytest.approx(gold[k], abs=1e-5), k
    assert qe._reinforce._reinforce_baseline == pytest.approx(gold["baseline_after"], abs=1e-5)
    for k in ("pg_loss", "qr_loss"):
        assert lv[k].grad.tolist() == pytest.approx(gold["grads"][k], abs=1e-6), k
    # second call: against the oracle (itself pinned by the reference's two-call known answers)
    cpu = {k: torch.tensor(v, requires_grad=True) for k, v in known["fixed_losses"].items()}
    r = elbo_oracle.Reinforce(0.99)
    elbo_oracle.question_coding_elbo(r, 0.1, cpu["pg_loss"], cpu["qr_loss"], cpu["prior_loss"])
    want = elbo_oracle.question_coding_elbo(r, 0.1, cpu["pg_loss"], cpu["qr_loss"], cpu["prior_loss"])
    for v in lv.values():
        v.grad = None
    out = qe.combine(lv["pg_loss"], lv["qr_loss"], lv["prior_loss"].detach())
    (-out["elbo"]).backward()
    (-want["elbo"]).backward()
    for k in out:
        assert float(out[k]) == pytest.approx(float(want[k]), abs=1e-5), k
    assert qe._reinforce._reinforce_baseline == pytest.approx(r.baseline, abs=1e-5)
    for k in ("pg_loss", "qr_loss"):
        torch.testing.assert_close(lv[k].grad.cpu(), cpu[k].grad, rtol=1e-5, atol=1e-6)
