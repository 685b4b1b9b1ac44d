"""One joint-training iteration (PG + QR + prior + NMN + REINFORCE/ELBO + clamp + Adam) on the MI355X
against the CPU oracle, replaying the device's sampled programs through the oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nhwc_images", [False, True])
def test_joint_training_step_matches_oracle(nhwc_images):
    """``nhwc_images``: the batch's features as the ingest kernel hands them over (a channels_last tensor, used in
    place by the stem; the joint step then takes the unsupervised rows out of it)."""
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
    if nhwc_images:
        dbatch["image"] = dbatch["image"].contiguous(memory_format=torch.channels_last)
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


def _question_coding_case(n, supervision, seed):
    from oracle.train_oracle import OracleQuestionCodingTrainer
    from probnmn.data.synthetic import synthetic_batch
    from probnmn.models import ProgramGenerator, ProgramPrior, QuestionReconstructor
    from probnmn.trainers.joint_training import QuestionCodingStep
    from probnmn.vocabulary import Vocabulary

    dev = torch.device("cuda:0")
    vocab = Vocabulary.clevr()
    torch.manual_seed(seed)
    pg, qr, prior = ProgramGenerator(vocab), QuestionReconstructor(vocab), ProgramPrior(vocab, hidden_size=256)
    sds = [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in (pg, qr, prior)]
    sds[2].pop("_output_layer.weight")
    batch = synthetic_batch(vocab, n, seed=seed + 3, with_image=False)
    supervision(batch["supervision"])
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


def test_question_coding_step_matches_oracle():
    def supervision(s):
        s[:3] = 1
        s[3:6] = 0
    _question_coding_case(14, supervision, 1)


def test_question_coding_step_matches_oracle_at_72_rows_with_an_uneven_split():
    """The question_coding iteration (question_coding_trainer.py:109-168) at a size where the batched passes of this
    build matter: 72 rows -- more than one 64-row launch of the paired decoders, five 16-row tiles per multi-CU
    cluster -- of which only 17 carry program supervision (the two cross entropies average over 17 rows, the REINFORCE
    terms over 55; mean-of-means would be wrong), against OracleQuestionCodingTrainer replaying the device's samples."""
    def supervision(s):
        s[:] = 0
        s[torch.tensor([0, 3, 4, 9, 10, 11, 20, 21, 33, 40, 41, 42, 43, 57, 60, 66, 71])] = 1
    _question_coding_case(72, supervision, 7)


def test_joint_training_step_at_config5_shapes_matches_oracle():
    """BASELINE configs[4] as ONE composed iteration: 28x28 feature maps through the banded conv kernels, a generator
    that decodes 40 steps (supervised programs of up to 40 tokens, synthetic deep shapes), reconstructor, prior,
    REINFORCE-ELBO, clamp, Adam -- against OracleJointTrainer(pg_steps=40) replaying the device's samples.  (Reduced
    classifier widths: the 200704 -> 1024 layer is covered at full size by tests/test_nmn_gpu.py.)"""
    from oracle.train_oracle import OracleJointTrainer
    from probnmn.data.synthetic import synthetic_batch
    from probnmn.models import NeuralModuleNetwork, ProgramGenerator, ProgramPrior, QuestionReconstructor
    from probnmn.trainers.joint_training import JointTrainingStep
    from probnmn.vocabulary import Vocabulary

    dev = torch.device("cuda:0")
    vocab = Vocabulary.clevr()
    size = (1024, 28, 28)
    torch.manual_seed(5)
    pg, qr = ProgramGenerator(vocab, max_decoding_steps=40), QuestionReconstructor(vocab)
    prior = ProgramPrior(vocab, hidden_size=256)
    nmn = NeuralModuleNetwork(vocab, image_feature_size=size, class_projection_channels=128, classifier_linear_size=64)
    sds = [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in (pg, qr, prior, nmn)]
    sds[2].pop("_output_layer.weight")
    batch = synthetic_batch(vocab, 10, image_feature_size=size, seed=6, deep=True)
    assert int((batch["program"] != 0).sum(1).max()) > 26  # longer than the 14x14 configurations' 26 steps
    batch["supervision"] = torch.tensor([1, 0, 0, 1, 0, 1, 0, 0, 1, 0])
    for m in (pg, qr, prior, nmn):
        m.to(dev)
    # teach the generator the deep programs for a few iterations so that its SAMPLES contain valid long programs
    from probnmn.optim import ClampAdam

    opt = ClampAdam(list(pg.parameters()), lr=2e-3, clamp=5.0)
    q_dev, p_dev = batch["question"].to(dev), batch["program"].to(dev)
    for _ in range(150):
        opt.zero_grad()
        pg(q_dev, p_dev, decoding_strategy="sampling")["loss"].mean().backward()
        opt.step()
    sds[0] = {k: v.detach().cpu().clone() for k, v in pg.state_dict().items()}
    lr = 1e-4
    hyper = dict(objective="ours", alpha=100.0, beta=0.1, gamma=1.0, delta=0.99, lr=lr)
    step = JointTrainingStep(pg, qr, prior, nmn, **hyper)
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    dbatch["supervision"] = batch["supervision"]
    out = step.step(dbatch)
    z = out["programs"].detach().cpu()
    torch.cuda.synchronize()
    assert z.shape[1] == 40
    ref = OracleJointTrainer(*sds, vocab.get_index_to_token_vocabulary("programs"), pg_steps=40, **hyper)
    ref_out = ref.step(batch, forced_programs=z)
    assert torch.equal(ref_out["programs"], z)
    valid = nmn.engine.compiler.compile_batch(z)
    assert sum(1 for c in valid if c.valid) >= 3, "the fitted generator should sample valid deep programs"
    assert float(out["loss"]["nmn"]) == pytest.approx(float(ref_out["nmn_loss"]), rel=1e-4, abs=1e-4)
    for k in ("elbo", "kl_divergence", "reconstruction_likelihood", "reinforce_reward"):
        assert float(out["elbo"][k]) == pytest.approx(float(ref_out["elbo"][k]), rel=1e-4, abs=1e-4), k
    assert float(out["objective"]) == pytest.approx(float(ref_out["objective"]), rel=1e-4, abs=1e-3)
    for key, model in (("pg", pg), ("qr", qr), ("nmn", nmn)):
        for name, p in model.named_parameters():
            g_ref = ref_out["grads"][key][name]
            if g_ref is None or p.grad is None:
                continue
            got = p.grad.detach().cpu().clamp(-5, 5)
            # flip-proof l2 bar for the trunk (a gate within round-off of zero reroutes one example's gradient:
            # tests/test_nmn_per_module_gpu.py holds the tight per-kind bar), max-norm bar for the seq2seq models
            if key == "nmn":
                assert float((got - g_ref).norm()) <= 5e-2 * float(g_ref.norm()) + 1e-9, (key, name)
            else:
                assert float((got - g_ref).abs().max()) / (float(g_ref.abs().max()) + 1e-12) < 5e-3, (key, name)


def test_joint_training_step_at_config5_shapes_with_the_full_classifier():
    """The config-5 joint iteration with the reference's FULL classifier widths (1024 projection channels, the
    200 704 -> 1024 fully connected layer: 822 MB of weights, and as much again in the oracle) at 4 rows: what the
    reduced-width test above leaves out.  The bars are the composed test's."""
    from oracle.train_oracle import OracleJointTrainer
    from probnmn.data.synthetic import synthetic_batch
    from probnmn.models import NeuralModuleNetwork, ProgramGenerator, ProgramPrior, QuestionReconstructor
    from probnmn.optim import ClampAdam
    from probnmn.trainers.joint_training import JointTrainingStep
    from probnmn.vocabulary import Vocabulary

    dev = torch.device("cuda:0")
    vocab = Vocabulary.clevr()
    size = (1024, 28, 28)
    torch.manual_seed(11)
    pg, qr = ProgramGenerator(vocab, max_decoding_steps=40), QuestionReconstructor(vocab)
    prior = ProgramPrior(vocab, hidden_size=256)
    nmn = NeuralModuleNetwork(vocab, image_feature_size=size)  # class_projection_channels = classifier_linear_size = 1024
    assert nmn.classifier[4].weight.shape == (1024, 1024 * 14 * 14)
    batch = synthetic_batch(vocab, 4, image_feature_size=size, seed=12, deep=True)
    batch["supervision"] = torch.tensor([1, 0, 0, 1])
    for m in (pg, qr, prior, nmn):
        m.to(dev)
    opt = ClampAdam(list(pg.parameters()), lr=2e-3, clamp=5.0)
    q_dev, p_dev = batch["question"].to(dev), batch["program"].to(dev)
    for _ in range(150):  # (so that the two sampled programs are valid ones: the NMN then runs on them)
        opt.zero_grad()
        pg(q_dev, p_dev, decoding_strategy="sampling")["loss"].mean().backward()
        opt.step()
    sds = [{k: v.detach().cpu().clone() for k, v in m.state_dict().items()} for m in (pg, qr, prior, nmn)]
    sds[2].pop("_output_layer.weight")
    lr = 1e-4
    hyper = dict(objective="ours", alpha=100.0, beta=0.1, gamma=1.0, delta=0.99, lr=lr)
    step = JointTrainingStep(pg, qr, prior, nmn, **hyper)
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    dbatch["supervision"] = batch["supervision"]
    out = step.step(dbatch)
    z = out["programs"].detach().cpu()
    torch.cuda.synchronize()
    ref = OracleJointTrainer(*sds, vocab.get_index_to_token_vocabulary("programs"), pg_steps=40, **hyper)
    ref_out = ref.step(batch, forced_programs=z)
    assert torch.equal(ref_out["programs"], z)
    assert float(out["loss"]["nmn"]) == pytest.approx(float(ref_out["nmn_loss"]), rel=1e-4, abs=1e-4)
    for k in ("elbo", "kl_divergence", "reconstruction_likelihood", "reinforce_reward"):
        assert float(out["elbo"][k]) == pytest.approx(float(ref_out["elbo"][k]), rel=1e-4, abs=1e-4), k
    assert float(out["objective"]) == pytest.approx(float(ref_out["objective"]), rel=1e-4, abs=1e-3)
    for key, model in (("pg", pg), ("qr", qr), ("nmn", nmn)):
        for name, p in model.named_parameters():
            g_ref = ref_out["grads"][key][name]
            if g_ref is None or p.grad is None:
                continue
            got = p.grad.detach().cpu().clamp(-5, 5)
            if key == "nmn":
                assert float((got - g_ref).norm()) <= 5e-2 * float(g_ref.norm()) + 1e-9, (key, name)
            else:
                assert float((got - g_ref).abs().max()) / (float(g_ref.abs().max()) + 1e-12) < 5e-3, (key, name)


def test_grouped_decoder_backward_is_the_pair_and_single_backward():
    """The backward passes of the generator's two decodes and the reconstructor's in ONE launch (round 5,
    ``_AttnLSTMDecoderGroup`` / ``pnmn_attn_lstm_bwd_multi_group3``) against pair + single launches: same kernels' bodies,
    same operands -- the decoders' own gradients are bit-identical and everything else agrees to the order of the embedding gradient's
    atomic adds, for a question-coding iteration (no NMN: nothing
    atomic in the step) at two batch sizes, one of them with a shard the three passes do not fit the chip for together."""
    from probnmn.data.synthetic import synthetic_batch
    from probnmn.models import ProgramGenerator, ProgramPrior, QuestionReconstructor
    from probnmn.trainers.joint_training import QuestionCodingStep
    from probnmn.vocabulary import Vocabulary

    dev = torch.device("cuda:0")
    vocab = Vocabulary.clevr()
    for n in (40, 600):
        grads, losses = {}, {}
        for grouped in (True, False):
            torch.manual_seed(11)
            pg, qr, prior = ProgramGenerator(vocab).to(dev), QuestionReconstructor(vocab).to(dev), ProgramPrior(vocab, hidden_size=256).to(dev)
            batch = synthetic_batch(vocab, n, seed=5, with_image=False)
            dbatch = {k: v.to(dev) for k, v in batch.items()}
            dbatch["supervision"] = batch["supervision"]
            step = QuestionCodingStep(pg, qr, prior, objective="ours", lr=0.0)  # (lr 0: the gradients stay to be read)
            step.group_decoder_backward = grouped
            torch.manual_seed(12)
            out = step.step(dbatch)
            torch.cuda.synchronize()
            losses[grouped] = float(out["objective"])
            grads[grouped] = {"%s.%s" % (k, name): p.grad.detach().clone() for k, m in (("pg", pg), ("qr", qr))
                              for name, p in m.named_parameters() if p.grad is not None}
        assert losses[True] == losses[False]
        assert set(grads[True]) == set(grads[False]) and len(grads[True]) > 20
        exact = 0
        for name in grads[True]:
            # (what passes through pnmn_embedding_grad -- per-token sums by global atomics -- differs in the last digits from
            # run to run; the decoders' own weights come straight out of the grouped launch and the GEMMs behind it)
            if "_decoder_cell.weight_hh" in name or "_output_projection_layer" in name:
                assert torch.equal(grads[True][name], grads[False][name]), (n, name)
                exact += 1
            else:
                scale = float(grads[False][name].abs().max())
                torch.testing.assert_close(grads[True][name], grads[False][name], rtol=0.0, atol=2e-6 * scale + 1e-12,
                                           msg=lambda m: "%s %s: %s" % (n, name, m))
        assert exact >= 6
