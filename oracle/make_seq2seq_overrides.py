"""Generate tests/golden/seq2seq_overrides.npz by EXECUTING the reference's own seq2seq lines (test infrastructure;
build container only, where /root/reference exists).

    python oracle/make_seq2seq_overrides.py

The reference files are loaded from where they lie -- probnmn/modules/seq2seq_base.py, probnmn/models/
{program_generator,question_reconstructor,program_prior}.py -- over oracle/allennlp_standin.py (AllenNLP 0.9.0 is
absent from this image; see that file for what is and is not pinned this way).  What runs is the reference's own
``forward`` / ``_forward_loop`` / ``_trim_predictions`` / ``_get_loss`` and ``ProgramPrior.forward``, at the
reference's dimensions (input 256, hidden 256, two layers, CLEVR-sized vocabularies), on inputs and weights from
oracle/detgen.py seeds recorded in the file.  ``torch.multinomial`` is replaced by an inverse-CDF draw from recorded
uniforms so that the sampled tokens are a function of the probabilities alone; the raw draws are stored.

Stored per model and case: per-row losses, (trimmed) predictions, raw draws, and for the training cases a digest of
every parameter gradient of ``loss.mean()`` (l2 norm, sum, 128 fixed entries).

  pg / qr   teacher   train mode, target tokens given (cross entropy; the draws only fill `predictions`)
            sample    train mode, no targets: free-running sampling, loss = -length-normalised log-probability
            greedy    eval mode, no targets
            validate  eval mode, target tokens given, greedy predictions
  prior     train mode: per-sequence cross entropy of the language model
"""
import importlib.util
import os
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden", "seq2seq_overrides.npz")
sys.path.insert(0, ROOT)

from oracle import allennlp_standin, detgen  # noqa: E402
from oracle.make_golden import _Vocab, namespaces  # noqa: E402

from oracle.seq2seq_fixture import SEEDS, grad_entries, inputs, weights  # noqa: E402


def grad_digest(out: dict, prefix: str, named_grads) -> None:
    for name, g in named_grads:
        flat = g.detach().reshape(-1).double().numpy()
        idx = grad_entries(name, flat.size)
        out["%s/grad/%s/norm" % (prefix, name)] = np.float64(np.sqrt((flat * flat).sum()))
        out["%s/grad/%s/sum" % (prefix, name)] = np.float64(flat.sum())
        out["%s/grad/%s/at" % (prefix, name)] = flat[idx].astype(np.float32)


class replayed_multinomial:
    """torch.multinomial(p, 1) -> inverse CDF of p at recorded uniforms (row-wise), draws remembered."""

    def __init__(self, seed):
        self.gen = detgen.rng(seed)
        self.draws = []

    def __enter__(self):
        self._orig = torch.multinomial

        def draw(probs, num_samples, *a, **k):
            assert num_samples == 1 and probs.dim() == 2
            p = probs.detach().double().numpy()
            cdf = np.cumsum(p, axis=1)
            u = self.gen.random(p.shape[0]) * cdf[:, -1]
            choice = np.array([min(int(np.searchsorted(cdf[i], u[i], side="right")), int(np.nonzero(p[i])[0][-1]))
                               for i in range(p.shape[0])])
            self.draws.append(choice.copy())
            return torch.from_numpy(choice).view(-1, 1)

        torch.multinomial = draw
        return self

    def __exit__(self, *exc):
        torch.multinomial = self._orig


def _load(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def reference_models():
    import types

    allennlp_standin.install(_Vocab)
    for name in ("probnmn", "probnmn.modules", "probnmn.models", "probnmn.utils"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["probnmn.config"] = types.ModuleType("probnmn.config")
    sys.modules["probnmn.config"].Config = object
    sys.modules["probnmn.utils.metrics"] = types.ModuleType("probnmn.utils.metrics")
    sys.modules["probnmn.utils.metrics"].SemanticQuestionReconstructionAccuracy = allennlp_standin._Recorder
    _load("probnmn.modules.seq2seq_base", "probnmn/modules/seq2seq_base.py")
    pg = _load("probnmn.models.program_generator", "probnmn/models/program_generator.py").ProgramGenerator
    qr = _load("probnmn.models.question_reconstructor", "probnmn/models/question_reconstructor.py").QuestionReconstructor
    prior = _load("probnmn.models.program_prior", "probnmn/models/program_prior.py").ProgramPrior
    vocab = _Vocab(namespaces())
    return vocab, pg(vocab), qr(vocab), prior(vocab, hidden_size=256)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    vocab, pg, qr, prior = reference_models()
    q, p = inputs()
    assert vocab.get_vocab_size("questions") == 100 and vocab.get_vocab_size("programs") == 44
    out = {"meta/seeds": np.array([SEEDS[k] for k in ("pg", "qr", "prior", "inputs", "draws")]),
           "meta/questions": q.numpy(), "meta/programs": p.numpy()}
    for kind, model, src, tgt in (("pg", pg, q, p), ("qr", qr, p, q)):
        sd = weights(kind)
        missing = model.load_state_dict(sd, strict=True)  # the reference's own parameter names
        assert not missing.missing_keys and not missing.unexpected_keys
        for case, train, with_target, strategy in (("teacher", True, True, "sampling"), ("sample", True, False, "sampling"),
                                                   ("greedy", False, False, "greedy"), ("validate", False, True, "greedy")):
            model.train(train)
            model.zero_grad()
            with replayed_multinomial(SEEDS["draws"] + zlib.crc32((kind + case).encode()) % 1000) as rm:
                res = model(src, tgt if with_target else None, decoding_strategy=strategy)
            pre = "%s/%s" % (kind, case)
            out[pre + "/loss"] = res["loss"].detach().numpy().astype(np.float32)
            out[pre + "/predictions"] = res["predictions"].numpy()
            if rm.draws:
                out[pre + "/raw_draws"] = np.stack(rm.draws, 1)
            if train:
                res["loss"].mean().backward()
                grad_digest(out, pre, [(n, t.grad) for n, t in model.named_parameters()])
            print(pre, "loss", np.round(out[pre + "/loss"], 4).tolist())
    sd = weights("prior")
    sd["_output_layer.weight"] = sd["_embedder.token_embedder_programs.weight"]
    prior.load_state_dict(sd, strict=True)
    prior.train()
    with replayed_multinomial(SEEDS["draws"] + 7) as rm:
        res = prior(p)
    out["prior/train/loss"] = res["loss"].detach().numpy().astype(np.float32)
    out["prior/train/predictions"] = res["predictions"].numpy()
    res["loss"].mean().backward()
    # (the tied output layer is the same Parameter as the embedding: named_parameters lists it once)
    grad_digest(out, "prior/train", [(n, t.grad) for n, t in prior.named_parameters()])
    print("prior loss", np.round(out["prior/train/loss"], 4).tolist())
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
