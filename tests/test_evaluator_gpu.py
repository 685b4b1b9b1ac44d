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
