"""Trajectory parity (north_star: "match the reference's module_training and joint_training answer accuracy within
stated FP tolerance on identical inputs").  Every other HIP-vs-oracle test is ONE iteration; here both sides TRAIN from
the same weights on the same stream of fresh batches of a learnable synthetic task -- answers are a deterministic
function of the ground-truth program and the features -- and are then validated on a held-out batch, as the reference's
loop does (scripts/train.py:135-140 -> trainers/_trainer.py:135-151 every iteration, evaluators/_evaluator.py:67-115
at the checkpoints).

Stated tolerances ("Numerics"): per-iteration loss within LOSS_TOL relative (fp32 accumulation
order + the occasional hard-gate flip, fed back through Adam for N iterations); validation answer accuracy within ONE
example; prediction agreement >= 95 %."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

MODULE_ITERS, MODULE_BATCH, MODULE_LOSS_TOL = 30, 16, 2e-3
JOINT_ITERS, JOINT_BATCH, JOINT_TOL = 10, 12, 2e-3
HELD_OUT = 64


def learnable_batch(vocab, n, seed):
    """A synthetic CLEVR-shaped batch whose answer is a deterministic function of program and features: the program's
    token sum picks the base class, one coarse feature statistic (which half of the first 128 channels is brighter)
    the offset."""
    from probnmn.data.synthetic import synthetic_batch

    b = synthetic_batch(vocab, n, seed=seed)
    num_answers = vocab.get_vocab_size("answers") - 1
    bit = (b["image"][:, :64].mean(dim=(1, 2, 3)) > b["image"][:, 64:128].mean(dim=(1, 2, 3))).long()
    b["answer"] = (b["program"].sum(dim=1) * 2 + bit) % num_answers
    return b


def to_dev(batch, dev):
    out = {k: v.to(dev) for k, v in batch.items()}
    out["supervision"] = batch["supervision"]  # (host copy: drives the host-side split)
    return out


def test_module_training_trajectory_and_validation_match_oracle():
    from oracle import nmn_oracle
    from oracle.train_oracle import OracleModuleTrainer
    from probnmn.models.nmn import NeuralModuleNetwork
    from probnmn.trainers.module_training import ModuleTrainingStep
    from probnmn.vocabulary import Vocabulary

    dev = torch.device("cuda:0")
    vocab = Vocabulary.clevr()
    itos = vocab.get_index_to_token_vocabulary("programs")
    torch.manual_seed(0)
    net = NeuralModuleNetwork(vocab)
    cpu_sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net.to(dev)
    lr = 1e-4
    trainer = ModuleTrainingStep(net, lr=lr)
    ref = OracleModuleTrainer(cpu_sd, itos, lr=lr)
    got, want = [], []
    for it in range(MODULE_ITERS):
        batch = learnable_batch(vocab, MODULE_BATCH, seed=1000 + it)  # (a fresh batch every iteration)
        got.append(float(trainer.step(to_dev(batch, dev))["loss"]))
        want.append(float(ref.step(batch)["loss"]))
    got, want = np.array(got), np.array(want)
    rel = np.abs(got - want) / np.maximum(1.0, np.abs(want))
    print("module_training loss curves: first %.5f / %.5f, last %.5f / %.5f, worst relative gap %.2e at iteration %d"
          % (got[0], want[0], got[-1], want[-1], rel.max(), int(rel.argmax())))
    assert rel.max() <= MODULE_LOSS_TOL, (rel.max(), int(rel.argmax()))
    assert want[-5:].mean() < want[:5].mean(), "the task must be learnable: the oracle's loss did not fall"

    # validation on a held-out batch, ground-truth programs (module training validates the NMN alone)
    held = learnable_batch(vocab, HELD_OUT, seed=77)
    net.eval()
    with torch.no_grad():
        dheld = to_dev(held, dev)
        pred = net(dheld["image"], dheld["program"], dheld["answer"])["predictions"].cpu()
    net.train()
    with torch.no_grad():
        ref_pred = nmn_oracle.nmn_forward(ref.params, itos, held["image"], held["program"], held["answer"])["predictions"]
    acc, ref_acc = int((pred == held["answer"]).sum()), int((ref_pred == held["answer"]).sum())
    agree = float((pred == ref_pred).float().mean())
    print("module_training validation: %d / %d correct (oracle %d), prediction agreement %.3f" % (acc, HELD_OUT, ref_acc, agree))
    assert abs(acc - ref_acc) <= 1
    assert agree >= 0.95


def test_joint_training_trajectory_and_validation_match_oracle():
    from oracle import nmn_oracle, seq2seq_oracle as so
    from oracle.train_oracle import OracleJointTrainer
    from probnmn.evaluators import evaluate_answer_accuracy
    from probnmn.models import NeuralModuleNetwork, ProgramGenerator, ProgramPrior, QuestionReconstructor
    from probnmn.optim import ClampAdam
    from probnmn.trainers.joint_training import JointTrainingStep
    from probnmn.vocabulary import Vocabulary

    dev = torch.device("cuda:0")
    vocab = Vocabulary.clevr()
    itos = vocab.get_index_to_token_vocabulary("programs")
    torch.manual_seed(0)
    pg, qr = ProgramGenerator(vocab), QuestionReconstructor(vocab)
    prior, nmn = ProgramPrior(vocab, hidden_size=256), NeuralModuleNetwork(vocab)
    for m in (pg, qr, prior, nmn):
        m.to(dev)
    # a pool of questions the generator is first fitted on (supervised, device side only), so that its samples are
    # mostly valid programs and the NMN half of the joint iteration has work to do; both sides then start from that state
    pool = learnable_batch(vocab, 48, seed=500)
    dpool = to_dev(pool, dev)
    opt = ClampAdam(list(pg.parameters()), lr=2e-3, clamp=5.0)
    for _ in range(150):
        opt.zero_grad()
        pg(dpool["question"], dpool["program"], decoding_strategy="sampling")["loss"].mean().backward()
        opt.step()
    torch.cuda.synchronize()
    sds = [{k: v.detach().cpu().clone() for k, v in m.state_dict().items()} for m in (pg, qr, prior, nmn)]
    sds[2].pop("_output_layer.weight", None)

    hyper = dict(objective="ours", alpha=100.0, beta=0.1, gamma=1.0, delta=0.99, lr=1e-4)
    step = JointTrainingStep(pg, qr, prior, nmn, **hyper)
    ref = OracleJointTrainer(*sds, itos, **hyper)
    rng = np.random.Generator(np.random.Philox(9))
    gaps, valid = [], 0
    for it in range(JOINT_ITERS):
        rows = torch.from_numpy(rng.choice(48, JOINT_BATCH, replace=False))
        batch = {k: v[rows] for k, v in pool.items()}
        batch["supervision"] = torch.tensor([1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0])
        out = step.step(to_dev(batch, dev))
        z = out["programs"].detach().cpu()
        ref_out = ref.step(batch, forced_programs=z)  # (the oracle replays the device's samples)
        assert torch.equal(ref_out["programs"], z)
        for k in ("elbo", "kl_divergence", "reconstruction_likelihood", "reinforce_reward"):
            g, w = float(out["elbo"][k]), float(ref_out["elbo"][k])
            gaps.append(abs(g - w) / max(1.0, abs(w)))
        g, w = float(out["objective"]), float(ref_out["objective"])
        gaps.append(abs(g - w) / max(1.0, abs(w)))
        g, w = float(out["loss"]["nmn"]), float(ref_out["nmn_loss"])
        gaps.append(abs(g - w) / max(1.0, abs(w)))
    print("joint_training: worst relative gap of objective / elbo terms / nmn loss over %d iterations %.2e" % (JOINT_ITERS, max(gaps)))
    assert max(gaps) <= JOINT_TOL, max(gaps)

    # validation as the reference runs it: greedy ProgramGenerator -> NMN, answer accuracy
    held = learnable_batch(vocab, HELD_OUT, seed=78)
    held["question"] = pool["question"][torch.arange(HELD_OUT) % 48]  # (questions the generator knows: valid programs)
    held["program"] = pool["program"][torch.arange(HELD_OUT) % 48]
    metrics = evaluate_answer_accuracy(pg, nmn, [to_dev(held, dev)])
    with torch.no_grad():
        zr = so.seq2seq_forward(ref.pg, held["question"], held["program"], "greedy")["predictions"]
        ref_pred = nmn_oracle.nmn_forward(ref.nmn, itos, held["image"], zr, held["answer"])["predictions"]
    ref_acc = int((ref_pred == held["answer"]).sum())
    acc = metrics["nmn"]["answer_accuracy"] * HELD_OUT
    pg.eval(), nmn.eval()
    with torch.no_grad():
        dheld = to_dev(held, dev)
        zd = pg(dheld["question"], dheld["program"], decoding_strategy="greedy")["predictions"]
        pred = nmn(dheld["image"], zd, dheld["answer"])["predictions"].cpu()
    pg.train(), nmn.train()
    agree = float((pred == ref_pred).float().mean())
    same_programs = float((zd.cpu() == zr).all(dim=1).float().mean())
    print("joint_training validation: %.0f / %d correct (oracle %d), prediction agreement %.3f, identical greedy programs %.3f"
          % (acc, HELD_OUT, ref_acc, agree, same_programs))
    assert abs(acc - ref_acc) <= 1
    assert agree >= 0.95
