"""Validation answer accuracy as the reference measures it (reference:
probnmn/evaluators/_evaluator.py:67-115, module_training_evaluator.py:81-109,
joint_training_evaluator.py:74-103): models in eval mode, no gradients; the ProgramGenerator is called
WITH the ground-truth programs as targets and "greedy" decoding, so its predictions are the arg-max
of the teacher-forced distributions; the NMN answers on those; accuracy = #(prediction == answer) / N
(an invalid program predicts @@UNKNOWN@@ and never matches).  The loop stops when the batch counter
exceeds ``num_batches`` (so it sees ``num_batches + 2`` batches, like the reference)."""
from typing import Dict, Iterable, Optional

import torch


@torch.no_grad()
def evaluate_answer_accuracy(program_generator, nmn, batches: Iterable[Dict[str, torch.Tensor]],
                             num_batches: Optional[int] = None) -> Dict[str, Dict[str, float]]:
    was_training = (program_generator.training, nmn.training)
    program_generator.eval()
    nmn.eval()
    nmn.get_metrics(reset=True)
    try:
        for iteration, batch in enumerate(batches):
            pg_out = program_generator(batch["question"], batch["program"], decoding_strategy="greedy")
            nmn(batch["image"], pg_out["predictions"], batch["answer"])
            if num_batches is not None and iteration > num_batches:
                break
        return {"program_generator": program_generator.get_metrics(), "nmn": nmn.get_metrics()}
    finally:
        program_generator.train(was_training[0])
        nmn.train(was_training[1])
