"""One joint-training iteration (PG + QR + prior + NMN + REINFORCE/ELBO + clamp + Adam) on the MI355X
against the CPU oracle, replaying the device's sampled programs through the oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_joint_training_step_matches_oracle():
    from oracle.train_oracle import OracleJointTrainer
    from probnmn.data.synthetic import synthetic_batch
    from probnmn.models import NeuralModuleNetwork, ProgramGenerator, ProgramPrior, QuestionReconstructor
    from probnmn.trainers.joint_training import JointTrainingStep
    from probnmn.vocabulary import Vocabulary

    dev = torch.device("cuda:0")
    vocab = Vocabulary.clevr()
    torch.manual_seed(0)
    pg, qr = ProgramGenerator(vocab), QuestionReconstructor(vocab)
    prior, nmn = ProgramPrior(vocab, hidden_size=256), NeuralModuleNetwork(vocab)
    sds = [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in (pg, qr, prior, nmn)]
    sds[2].pop("_output_layer.weight")
    batch = synthetic_batch(vocab, 12, seed=3)
    batch["supervision"][:2] = 1
    batch["supervision"][2:4] = 0
    for m in (pg, qr, prior, nmn):
        m.to(dev)
    lr = 1e-4  # larger than the yml's 1e-6 so that one step moves parameters measurably
    step = JointTrainingStep(pg, qr, prior, nmn, objective="ours", alpha=100.0, beta=0.1, gamma=1.0, delta=0.99, lr=lr)
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    dbatch["supervision"] = batch["supervision"]  # host copy: no sync for the split
    out = step.step(dbatch)
    seen = {"z": out["programs"].detach().cpu()}  # the sampled programs the elbo saw
    torch.cuda.synchronize()

    ref = OracleJointTrainer(*sds, vocab.get_index_to_token_vocabulary("programs"), objective="ours", alpha=100.0,
                             beta=0.1, gamma=1.0, delta=0.99, lr=lr)
    ref_out = ref.step(batch, forced_programs=seen["z"])
    assert torch.equal(ref_out["programs"], seen["z"])
    assert float(out["loss"]["nmn"]) == pytest.approx(float(ref_out["nmn_loss"]), rel=1e-4, abs=1e-4)
    for k in ("elbo", "kl_divergence", "reconstruction_likelihood", "reinforce_reward"):
        assert float(out["elbo"][k]) == pytest.approx(float(ref_out["elbo"][k]), rel=1e-4, abs=1e-4), k
    assert float(out["objective"]) == pytest.approx(float(ref_out["objective"]), rel=1e-4, abs=1e-3)
    assert step.elbo._reinforce._reinforce_baseline == pytest.approx(ref_out["baseline"], rel=1e-4, abs=1e-4)
    # parameters after clamp + Adam: every element moved by at most lr and agrees with the oracle
    for model, ref_params in ((pg, ref.pg), (qr, ref.qr), (nmn, ref.nmn)):
        worst = 0.0
        for name, p in model.named_parameters():
            worst = max(worst, float((p.detach().cpu() - ref_params[name].detach()).abs().max()))
        assert worst <= 2.05 * lr, worst
    # gradients as the optimizer saw them (clamped to [-5, 5]); alpha = 100 pushes some past the clamp
    n_clamped = 0
    for key, model in (("pg", pg), ("qr", qr), ("nmn", nmn)):
        for name, p in model.named_parameters():
            g_ref = ref_out["grads"][key][name]
            if g_ref is None or p.grad is None:
                continue
            got = p.grad.detach().cpu().clamp(-5, 5)
            n_clamped += int((p.grad.detach().abs() > 5).sum())
            scale = float(g_ref.abs().max()) + 1e-12
            assert float((got - g_ref).abs().max()) / scale < 5e-3, (key, name)
    print("elements beyond the clamp:", n_clamped)
    # sampled programs that are invalid score the constant loss and get no NMN gradient
    assert out["elbo"]["elbo"].ndim == 0


def test_question_coding_step_matches_oracle():
    from oracle.train_oracle import OracleQuestionCodingTrainer
    from probnmn.data.synthetic import synthetic_batch
    from probnmn.models import ProgramGenerator, ProgramPrior, QuestionReconstructor
    from probnmn.trainers.joint_training import QuestionCodingStep
    from probnmn.vocabulary import Vocabulary

    dev = torch.device("cuda:0")
    vocab = Vocabulary.clevr()
    torch.manual_seed(1)
    pg, qr, prior = ProgramGenerator(vocab), QuestionReconstructor(vocab), ProgramPrior(vocab, hidden_size=256)
    sds = [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in (pg, qr, prior)]
    sds[2].pop("_output_layer.weight")
    batch = synthetic_batch(vocab, 14, seed=4, with_image=False)
    batch["supervision"][:3] = 1
    batch["supervision"][3:6] = 0
    for m in (pg, qr, prior):
        m.to(dev)
    step = QuestionCodingStep(pg, qr, prior, objective="ours", alpha=100.0, beta=0.1, delta=0.99, lr=1e-3)
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    dbatch["supervision"] = batch["supervision"]
    out = step.step(dbatch)
    seen = {"z": out["programs"].detach().cpu()}
    torch.cuda.synchronize()
    ref = OracleQuestionCodingTrainer(*sds, objective="ours", alpha=100.0, beta=0.1, delta=0.99, lr=1e-3)
    ref_out = ref.step(batch, forced_programs=seen["z"])
    for k in ("elbo", "kl_divergence", "reconstruction_likelihood", "reinforce_reward"):
        assert float(out["elbo"][k]) == pytest.approx(float(ref_out["elbo"][k]), rel=1e-4, abs=1e-4), k
    assert float(out["objective"]) == pytest.approx(float(ref_out["objective"]), rel=1e-4, abs=1e-2)
    for key, model in (("pg", pg), ("qr", qr)):
        for name, p in model.named_parameters():
            g_ref = ref_out["grads"][key][name]
            got = p.grad.detach().cpu().clamp(-5, 5)
            scale = float(g_ref.abs().max()) + 1e-12
            assert float((got - g_ref).abs().max()) / scale < 5e-3, (key, name)
