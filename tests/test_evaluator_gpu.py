"""Validation answer accuracy (teacher-forced greedy ProgramGenerator -> NMN) on the device equals the
same definition evaluated with the CPU oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_answer_accuracy_matches_oracle_definition():
    from oracle import nmn_oracle, seq2seq_oracle as so
    from probnmn.data.synthetic import synthetic_batch
    from probnmn.evaluators import evaluate_answer_accuracy
    from probnmn.models import NeuralModuleNetwork, ProgramGenerator
    from probnmn.vocabulary import Vocabulary

    dev = torch.device("cuda:0")
    vocab = Vocabulary.clevr()
    torch.manual_seed(0)
    pg, nmn = ProgramGenerator(vocab), NeuralModuleNetwork(vocab)
    pg_sd = {k: v.detach().clone() for k, v in pg.state_dict().items()}
    nmn_sd = {k: v.detach().clone() for k, v in nmn.state_dict().items()}
    batches = [synthetic_batch(vocab, 6, seed=s) for s in (1, 2, 3, 4)]
    pg.to(dev)
    nmn.to(dev)
    metrics = evaluate_answer_accuracy(pg, nmn, [{k: v.to(dev) for k, v in b.items()} for b in batches], num_batches=1)
    # num_batches = 1 -> the loop sees batches 0, 1, 2 (iteration > num_batches breaks after the third)
    correct = total = 0
    itos = vocab.get_index_to_token_vocabulary("programs")
    with torch.no_grad():
        for b in batches[:3]:
            z = so.seq2seq_forward(pg_sd, b["question"], b["program"], "greedy")["predictions"]
            out = nmn_oracle.nmn_forward(nmn_sd, itos, b["image"], z, b["answer"])
            correct += int((out["predictions"] == b["answer"]).sum())
            total += b["answer"].numel()
    assert metrics["nmn"]["answer_accuracy"] == pytest.approx(correct / total)
    assert pg.training and nmn.training  # modes restored


def test_phase_evaluators_and_inference():
    """the program-prior and question-coding evaluators (metrics = what the models accumulated
    over the batches the reference's loop sees) and scripts/inference.py's answer records."""
    from oracle import nmn_oracle, seq2seq_oracle as so
    from probnmn.data.synthetic import synthetic_batch
    from probnmn.evaluators import predict_answers, program_prior_evaluator, question_coding_evaluator
    from probnmn.models import NeuralModuleNetwork, ProgramGenerator, ProgramPrior, QuestionReconstructor
    from probnmn.vocabulary import Vocabulary

    dev = torch.device("cuda:0")
    vocab = Vocabulary.clevr()
    torch.manual_seed(0)
    pg, qr, prior = ProgramGenerator(vocab), QuestionReconstructor(vocab), ProgramPrior(vocab, hidden_size=256)
    nmn = NeuralModuleNetwork(vocab, class_projection_channels=128, classifier_linear_size=64)
    sds = {n: {k: v.detach().clone() for k, v in m.state_dict().items()} for n, m in (("pg", pg), ("qr", qr), ("prior", prior), ("nmn", nmn))}
    for m in (pg, qr, prior, nmn):
        m.to(dev)
    host = [synthetic_batch(vocab, 6, seed=s) for s in (11, 12, 13, 14)]
    batches = [{k: v.to(dev) for k, v in b.items()} for b in host]

    m = program_prior_evaluator(prior).evaluate(batches, num_batches=0)  # sees batches 0 and 1
    psd = {k: v for k, v in sds["prior"].items() if k != "_output_layer.weight"}
    want = sum(float(so.program_prior_loss(psd, b["program"]).mean()) for b in host[:2]) / 2
    assert m["program_prior"]["perplexity"] == pytest.approx(2 ** want, rel=1e-4)

    m = question_coding_evaluator(pg, qr).evaluate(batches)  # all four batches
    assert set(m) == {"program_generator", "question_reconstructor"}
    ce = [float(so.seq2seq_forward(sds["pg"], b["question"], b["program"], "greedy")["loss"].mean()) for b in host]
    assert m["program_generator"]["perplexity"] == pytest.approx(2 ** (sum(ce) / 4), rel=1e-4)
    for k in ("BLEU", "sequence_accuracy", "word_error_rate"):
        assert 0.0 <= m["question_reconstructor"][k] <= 1.0
    assert pg.training and qr.training and prior.training

    # inference: sampled programs -> NMN -> answer strings; the oracle replays the device's samples
    torch.manual_seed(5)
    records = predict_answers(pg, nmn, batches[:2], vocab)
    assert len(records) == 12 and [r["question_index"] for r in records] == list(range(12))
    torch.manual_seed(5)
    pg.eval()
    with torch.no_grad():
        z = [pg(b["question"])["predictions"].cpu() for b in batches[:2]]
    pg.train()
    itos = vocab.get_index_to_token_vocabulary("programs")
    k = 0
    for b, zz in zip(host[:2], z):
        out = nmn_oracle.nmn_forward(sds["nmn"], itos, b["image"], zz, None)
        for a in out["predictions"].tolist():
            assert records[k]["answer"] == vocab.get_token_from_index(a, "answers")
            k += 1


def test_bleu_counts_device_batches_later_and_the_same():
    """running_metrics.BLEU: device batches are copied out asynchronously and counted when the copy has landed (more batches
    than its ring of page-locked buffers, a host batch in between, get_metric at the end): the same corpus BLEU as counting
    every batch at once on the host."""
    import numpy as np
    from probnmn.running_metrics import BLEU

    rng = np.random.default_rng(0)
    dev = torch.device("cuda:0")
    a, b = BLEU(exclude_indices={0, 2, 3}), BLEU(exclude_indices={0, 2, 3})
    for k in range(7):
        pred = torch.from_numpy(rng.integers(0, 9, size=(5 + k % 2, 11)))
        gold = torch.from_numpy(rng.integers(0, 9, size=(5 + k % 2, 11)))
        a(pred, gold)
        if k == 3:
            b(pred, gold)  # (a host batch between device batches: counted behind the ones before it)
        else:
            b(pred.to(dev), gold.to(dev))
    assert b.get_metric(reset=True)["BLEU"] == a.get_metric(reset=True)["BLEU"]
    assert b.get_metric()["BLEU"] == a.get_metric()["BLEU"]  # (both empty again)
