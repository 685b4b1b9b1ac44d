"""Validation and inference loops over the MI355X models, as the reference runs them (reference:
probnmn/evaluators/_evaluator.py:67-115 and the four phase evaluators; scripts/inference.py:76-91).

Stand-alone layout only (when the models are grafted onto the reference's package, its own
``probnmn.evaluators`` keeps working on them).  Semantics kept:
  * models in ``eval()`` mode, no gradients, back to ``train()`` afterwards;
  * the loop stops when the batch counter EXCEEDS ``num_batches`` -- it sees ``num_batches + 2`` batches;
  * the ProgramGenerator is called WITH the ground-truth programs as targets and "greedy" decoding, so its
    predictions are the arg-max of the teacher-forced distributions (seq2seq_base.py:188-198), and the NMN
    answers on those; accuracy = #(prediction == answer) / N (an invalid program predicts @@UNKNOWN@@);
  * inference samples programs (``program_generator(question)`` with the default strategy, in eval mode)
    and lets the NMN answer without gold answers.
Batches are dicts of DEVICE tensors with the reference's keys (a ``PrefetchingLoader`` yields them)."""
from typing import Any, Callable, Dict, Iterable, List, Optional

import torch


class Evaluator:
    """``_Evaluator``: ``models`` name -> module (the trainer's dict, shared by reference), ``do_iteration(batch)``
    runs the phase's forward passes (which accumulate metrics inside the models)."""

    def __init__(self, models: Dict[str, torch.nn.Module], do_iteration: Callable[[Dict[str, torch.Tensor]], Any],
                 stages: Optional[tuple] = None):
        self.models = models
        self._do_iteration = do_iteration
        # (queue, finish): an iteration cut in two so that batch i + 1 is QUEUED before batch i is FINISHED -- same batches,
        # same order per model; what it buys is that the host's wait for a device result of batch i (the NMN needs the
        # generator's programs on the host) is spent with batch i + 1's kernels already in the queue
        self._stages = stages

    @torch.no_grad()
    def evaluate(self, batches: Iterable[Dict[str, torch.Tensor]], num_batches: Optional[int] = None) -> Dict[str, Dict[str, float]]:
        was_training = {k: m.training for k, m in self.models.items()}
        for m in self.models.values():
            m.eval()
            if hasattr(m, "get_metrics"):
                m.get_metrics(reset=True)
        try:
            if self._stages is None:
                for iteration, batch in enumerate(batches):
                    self._do_iteration(batch)
                    if num_batches is not None and iteration > num_batches:
                        break
            else:
                queue, finish = self._stages
                pending = None
                for iteration, batch in enumerate(batches):
                    queued = queue(batch, iteration)
                    if pending is not None:
                        finish(*pending)
                    pending = (batch, queued)
                    if num_batches is not None and iteration > num_batches:
                        break
                if pending is not None:
                    finish(*pending)
            return {k: m.get_metrics() for k, m in self.models.items() if hasattr(m, "get_metrics")}
        finally:
            for k, m in self.models.items():
                m.train(was_training[k])


def program_prior_evaluator(program_prior) -> Evaluator:
    """program_prior_evaluator.py: perplexity of the prior on validation programs."""
    return Evaluator({"program_prior": program_prior}, lambda b: program_prior(b["program"]))


def question_coding_evaluator(program_generator, question_reconstructor) -> Evaluator:
    """question_coding_evaluator.py:150-160: both models teacher-forced, "greedy"."""
    def it(b):
        return {"program_generator": program_generator(b["question"], b["program"], decoding_strategy="greedy"),
                "question_reconstructor": question_reconstructor(b["program"], b["question"], decoding_strategy="greedy")}
    return Evaluator({"program_generator": program_generator, "question_reconstructor": question_reconstructor}, it)


def answering_evaluator(program_generator, nmn) -> Evaluator:
    """module_training_evaluator.py:81-109 / joint_training_evaluator.py:74-103."""
    def it(b):
        pg_out = program_generator(b["question"], b["program"], decoding_strategy="greedy")
        return {"program_generator": pg_out, "nmn": nmn(b["image"], pg_out["predictions"], b["answer"])}

    # On the device the NMN's launch schedule depends on the programs, which it therefore reads back to the host
    # (models/nmn.py forward_trunk; the reference reads them back per example, nmn.py:203).  One batch at a time that is
    # generator -> wait -> plan -> NMN -> generator ...: the GPU idles while the host plans and the host idles while the
    # generator decodes.  Two stages instead: queue = generator pass + an asynchronous copy of the predictions into
    # one of two page-locked buffers; finish = wait for THAT copy, then the NMN on the host-side programs.
    pinned: Dict[Any, torch.Tensor] = {}

    def queue(b, iteration):
        pg_out = program_generator(b["question"], b["program"], decoding_strategy="greedy")
        pred = pg_out["predictions"]
        if not (pred.is_cuda and b["image"].is_cuda):
            return pg_out, None, None
        key = (iteration & 1, tuple(pred.shape), pred.dtype)
        if key not in pinned:
            pinned[key] = torch.empty(pred.shape, dtype=pred.dtype, pin_memory=True)
        host = pinned[key]
        host.copy_(pred, non_blocking=True)
        copied = torch.cuda.Event()
        copied.record()
        return pg_out, host, copied

    def finish(b, queued):
        pg_out, host, copied = queued
        if host is None:
            return nmn(b["image"], pg_out["predictions"], b["answer"])
        copied.synchronize()
        # (the whole NMN pass here, stem included: the engine has ONE activation arena, and a stem launched for batch i + 1
        # would overwrite batch i's before its module programs have run.  The trunk on the process's trunk stream, beside
        # batch i + 1's generator pass, measured the same: 4.23 / 4.61 against 4.42 / 4.44 ms per 256-question batch)
        return nmn(b["image"], host, b["answer"])

    return Evaluator({"program_generator": program_generator, "nmn": nmn}, it, stages=(queue, finish))


def evaluate_answer_accuracy(program_generator, nmn, batches: Iterable[Dict[str, torch.Tensor]],
                             num_batches: Optional[int] = None) -> Dict[str, Dict[str, float]]:
    """Validation answer accuracy (the metric joint / module training select checkpoints on)."""
    return answering_evaluator(program_generator, nmn).evaluate(batches, num_batches)


@torch.no_grad()
def predict_answers(program_generator, nmn, batches: Iterable[Dict[str, torch.Tensor]], vocabulary) -> List[Dict[str, Any]]:
    """scripts/inference.py:76-91: sampled programs -> NMN -> answer strings, one record per question
    (``question_index`` from the batch when present, else a running index)."""
    was_training = (program_generator.training, nmn.training)
    program_generator.eval()
    nmn.eval()
    records: List[Dict[str, Any]] = []
    try:
        # (as in answering_evaluator: batch i + 1's generator pass is queued before batch i's programs are awaited on the host)
        pinned: Dict[Any, torch.Tensor] = {}

        def queue(batch, iteration):
            programs = program_generator(batch["question"])["predictions"]
            if not (programs.is_cuda and batch["image"].is_cuda):
                return programs, None
            key = (iteration & 1, tuple(programs.shape), programs.dtype)
            if key not in pinned:
                pinned[key] = torch.empty(programs.shape, dtype=programs.dtype, pin_memory=True)
            pinned[key].copy_(programs, non_blocking=True)
            copied = torch.cuda.Event()
            copied.record()
            return pinned[key], copied

        def finish(batch, queued):
            programs, copied = queued
            if copied is not None:
                copied.synchronize()
            answers = nmn(batch["image"], programs)["predictions"].cpu().tolist()
            index = batch["question_index"].cpu().tolist() if "question_index" in batch else range(len(records), len(records) + len(answers))
            for qi, a in zip(index, answers):
                records.append({"question_index": int(qi), "answer": vocabulary.get_token_from_index(int(a), namespace="answers")})

        pending = None
        for iteration, batch in enumerate(batches):
            queued = queue(batch, iteration)
            if pending is not None:
                finish(*pending)
            pending = (batch, queued)
        if pending is not None:
            finish(*pending)
        return records
    finally:
        program_generator.train(was_training[0])
        nmn.train(was_training[1])
