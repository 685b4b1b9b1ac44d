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


@pytest.mark.parametrize("n,m,with_nmn,weighted", [(13, 7, True, False), (13, 7, False, True), (5, 0, True, True), (0, 9, False, False),
                                                   (300, 212, True, True)])
def test_fused_objective_equals_the_chain_of_torch_ops(n, m, with_nmn, weighted):
    """pnmn_joint_objective (one launch forward, one multiply backward) against the reference's arithmetic written out
    with torch ops (question_coding_trainer.py:128-165 / joint_training_trainer.py:150-191 over elbo.py:28-89,150-160,
    253-270): statistics, objective, every per-row gradient and the moving-baseline update -- incl. a batch without
    supervised rows, one without sampled rows, and data-parallel loss weights."""
    from probnmn.modules.elbo import JointTrainingElbo

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n * 31 + m)
    rows = lambda k: (torch.rand(k, generator=g) * 3 + 0.1).to(dev).requires_grad_(True)  # noqa: E731
    pg, prior, nmn, pg_sup = rows(n), rows(n).detach(), rows(n) if with_nmn else None, rows(m)
    qr = rows(n + m)
    alpha, beta, gamma, decay, b0 = 100.0, 0.1, 1.5 if with_nmn else 0.0, 0.9, 0.37
    w_u = torch.tensor(0.8, device=dev) if weighted else 1.0
    w_s = torch.tensor(1.3, device=dev) if weighted else 1.0
    elbo = JointTrainingElbo(None, None, None, None, beta=beta, gamma=gamma, baseline_decay=decay)
    elbo._reinforce._baseline = torch.tensor(b0, device=dev)
    J, stats = elbo.objective(pg if n else None, qr, prior if n else None, nmn if n else None, pg_sup if m else None,
                              w_u, w_s, alpha, gamma, n, m)
    J.backward()
    got = {"pg": pg.grad, "qr": qr.grad, "nmn": None if nmn is None else nmn.grad, "pg_sup": pg_sup.grad}
    new_baseline = float(elbo._reinforce._baseline)

    # ---- the same with torch ops
    pg2, qr2, pgs2 = pg.detach().clone().requires_grad_(True), qr.detach().clone().requires_grad_(True), pg_sup.detach().clone().requires_grad_(True)
    nmn2 = None if nmn is None else nmn.detach().clone().requires_grad_(True)
    want = torch.zeros((), device=dev)
    if n:
        logq, rec = -pg2, -qr2[:n]
        R = (rec + beta * (-prior) - beta * logq + (gamma * (-nmn2) if nmn2 is not None else 0.0)).detach()
        c = R - b0
        kl = logq * c - beta * logq
        e = (rec - kl).mean()
        want = want + w_u * ((gamma * nmn2.mean() if nmn2 is not None else 0.0) - e)
        assert float(stats["elbo"]) == pytest.approx(float(e), rel=1e-5, abs=1e-5)
        assert float(stats["reinforce_reward"]) == pytest.approx(float(R.mean()), rel=1e-5, abs=1e-5)
        assert float(stats["kl_divergence"]) == pytest.approx(float(kl.mean()), rel=1e-5, abs=1e-5)
        assert new_baseline == pytest.approx(b0 + decay * float(c.mean()), rel=1e-5, abs=1e-6)
    else:
        assert new_baseline == pytest.approx(b0)
    if m:
        want = want + w_s * alpha * (pgs2.mean() + qr2[n:].mean())
        assert float(stats["program_generation_gt"]) == pytest.approx(float(pgs2.mean()), rel=1e-5)
        assert float(stats["question_reconstruction_gt"]) == pytest.approx(float(qr2[n:].mean()), rel=1e-5)
    want.backward()
    assert float(J) == pytest.approx(float(want), rel=1e-5, abs=1e-4)
    for name, a, b in (("pg", got["pg"], pg2.grad), ("qr", got["qr"], qr2.grad), ("pg_sup", got["pg_sup"], pgs2.grad),
                       ("nmn", got["nmn"], None if nmn2 is None else nmn2.grad)):
        if b is None:
            assert a is None or float(a.abs().max()) == 0.0, name
        elif a is None:
            assert float(b.abs().max()) == 0.0, name
        else:
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6, msg=lambda msg, name=name: "%s: %s" % (name, msg))
