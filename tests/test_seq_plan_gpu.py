"""The seq2seq launch plan (probnmn.runtime.seq_plan) against the eager passes it replaces: the same iteration from the same
weights and the same torch seed through ``use_plan = True`` and ``False`` -- identical sampled programs, per-row losses and
objective to fp32 round-off, every parameter gradient within 2e-4 of its largest entry (the recurrent kernels, losses and
samplers are the same code; the products over all time steps are pnmn_gemm on one side, hipBLASLt on the other), and the
same parameters after two optimiser steps."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _models(seed, dev, nmn=False):
    from probnmn.models import NeuralModuleNetwork, ProgramGenerator, ProgramPrior, QuestionReconstructor
    from probnmn.vocabulary import Vocabulary

    vocab = Vocabulary.clevr()
    torch.manual_seed(seed)
    ms = [ProgramGenerator(vocab), QuestionReconstructor(vocab), ProgramPrior(vocab, hidden_size=256)]
    if nmn:
        ms.append(NeuralModuleNetwork(vocab))
    return vocab, ms


def _clone_into(dst, src):
    for d, s in zip(dst, src):
        d.load_state_dict(s.state_dict())


def _grads(models):
    return {"%d.%s" % (i, n): p.grad.detach().clone() for i, m in enumerate(models) for n, p in m.named_parameters()
            if p.grad is not None}


@pytest.mark.parametrize("n,sup", [(24, 7), (130, 64), (16, 15)])
def test_question_coding_plan_equals_eager(n, sup):
    from probnmn.data.synthetic import synthetic_batch
    from probnmn.trainers.joint_training import QuestionCodingStep

    dev = torch.device("cuda:0")
    vocab, a = _models(1, dev)
    _, b = _models(2, dev)
    _clone_into(b, a)
    batch = synthetic_batch(vocab, n, seed=n, with_image=False)
    batch["supervision"][:] = 0
    batch["supervision"][:sup] = 1
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    dbatch["supervision"] = batch["supervision"]
    outs, grads, steps = [], [], []
    for models, plan in ((a, True), (b, False)):
        for m in models:
            m.to(dev)
        step = QuestionCodingStep(*models, objective="ours", alpha=100.0, beta=0.1, delta=0.99, lr=0.0)
        step.use_plan = plan
        torch.manual_seed(77)
        outs.append(step.step(dbatch))
        grads.append(_grads(models[:2]))
        steps.append(step)
    assert steps[0].__dict__.get("_plans") and all(p is not False for p in steps[0]._plans.values())  # (the plan DID run)
    assert not steps[1].__dict__.get("_plans")
    assert torch.equal(outs[0]["programs"], outs[1]["programs"])
    for k in ("elbo", "kl_divergence", "reconstruction_likelihood", "reinforce_reward"):
        assert float(outs[0]["elbo"][k]) == pytest.approx(float(outs[1]["elbo"][k]), rel=2e-5, abs=2e-5), k
    for k in ("program_generation_gt", "question_reconstruction_gt"):
        assert float(outs[0]["loss"][k]) == pytest.approx(float(outs[1]["loss"][k]), rel=2e-5, abs=2e-5), k
    assert float(outs[0]["objective"]) == pytest.approx(float(outs[1]["objective"]), rel=2e-5, abs=1e-4)
    assert sorted(grads[0]) == sorted(grads[1])
    worst = {}
    for name in grads[0]:
        scale = float(grads[1][name].abs().max()) + 1e-12
        worst[name] = float((grads[0][name] - grads[1][name]).abs().max()) / scale
    bad = {k: v for k, v in worst.items() if v > 2e-4}
    assert not bad, bad
    # a second iteration: the plan is REPLAYED (same workspace, derived parameters refreshed).  Both trainers step with a
    # learning rate of zero, so the second iteration starts from identical weights again and the comparison stays sharp
    # (after a real Adam step elements with ~zero gradient differ by up to 2 lr between any two fp32 evaluation orders).
    second = []
    for step, models in zip(steps, (a, b)):
        torch.manual_seed(78)
        out2 = step.step(dbatch)
        second.append((out2, _grads(models[:2])))
    torch.cuda.synchronize()
    assert torch.equal(second[0][0]["programs"], second[1][0]["programs"])
    assert not torch.equal(second[0][0]["programs"], outs[0]["programs"])  # (another seed: other samples)
    assert float(second[0][0]["objective"]) == pytest.approx(float(second[1][0]["objective"]), rel=2e-5, abs=1e-4)
    for name in second[0][1]:
        scale = float(second[1][1][name].abs().max()) + 1e-12
        assert float((second[0][1][name] - second[1][1][name]).abs().max()) / scale < 2e-4, name


def test_joint_plan_equals_eager():
    from probnmn.data.synthetic import synthetic_batch
    from probnmn.trainers.joint_training import JointTrainingStep

    dev = torch.device("cuda:0")
    vocab, a = _models(3, dev, nmn=True)
    _, b = _models(4, dev, nmn=True)
    _clone_into(b, a)
    batch = synthetic_batch(vocab, 20, seed=5)
    batch["supervision"][:] = 0
    batch["supervision"][:9] = 1
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    dbatch["supervision"] = batch["supervision"]
    outs, grads = [], []
    for models, plan in ((a, True), (b, False)):
        for m in models:
            m.to(dev)
        step = JointTrainingStep(*models, objective="ours", alpha=100.0, beta=0.1, gamma=1.0, delta=0.99, lr=1e-5)
        step.use_plan = plan
        torch.manual_seed(5)
        outs.append(step.step(dbatch))
        torch.cuda.synchronize()
        grads.append(_grads(models[:2]))
        step.close()
    assert torch.equal(outs[0]["programs"], outs[1]["programs"])
    assert float(outs[0]["objective"]) == pytest.approx(float(outs[1]["objective"]), rel=2e-5, abs=1e-4)
    assert float(outs[0]["loss"]["nmn"]) == pytest.approx(float(outs[1]["loss"]["nmn"]), rel=1e-5, abs=1e-5)
    for name in grads[0]:
        scale = float(grads[1][name].abs().max()) + 1e-12
        assert float((grads[0][name] - grads[1][name]).abs().max()) / scale < 2e-4, name
