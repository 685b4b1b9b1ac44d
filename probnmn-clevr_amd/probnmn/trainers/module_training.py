"""One module-training iteration on one GPU (reference: probnmn/trainers/_trainer.py:135-151 and
module_training_trainer.py:88-98): zero_grad -> NMN forward on the given programs -> mean loss ->
backward -> clamp to [-5, 5] -> Adam.  Programs come from the batch (ground-truth or pre-sampled)
or, when a frozen ``program_generator`` is supplied, are sampled from it as the reference does."""
from typing import Any, Dict, Optional

import torch

from probnmn import parallel
from probnmn.optim import ClampAdam
from ._base import StepBase


class ModuleTrainingStep(StepBase):
    def __init__(self, nmn, lr: float = 1e-4, weight_decay: float = 0.0, program_generator=None,
                 report_metrics: bool = False, lr_gamma: float = 0.5, lr_patience: int = 1000000):
        self.nmn = nmn
        self.program_generator = program_generator
        self.report_metrics = report_metrics
        arena = nmn.engine.ensure_arena()
        nmn.engine.direct_grads = True  # gradients stay in the arena; the optimizer reads them there
        # this iteration has the chip to itself: no CU budget left over from a joint step that ran the same network's
        # trunk beside its seq2seq passes (bench.py's r03c record: 12.7 -> 14.1 ms with a stale budget of 192)
        nmn.engine.conv_cus = nmn.engine.wgrad_cus = 0
        self.optimizer = ClampAdam(nmn.parameters(), arenas=[arena], lr=lr, weight_decay=weight_decay, clamp=5.0)
        # data parallel: the big loose FC gradient starts its all-reduce while the trunk is still in backward
        big = [p for p in self.optimizer.loose if p.numel() >= (1 << 20)]
        self._early = parallel.early_reducer_for(big, [nmn.engine])
        self.models = {"nmn": nmn}  # (the frozen program generator is not checkpointed by this phase)
        self._init_schedule(lr_gamma, lr_patience)
        self.iteration = 0

    def step(self, batch: Dict[str, torch.Tensor]) -> Dict[str, Any]:
        self.optimizer.zero_grad()
        if self.program_generator is not None:
            with torch.no_grad():
                programs = self.program_generator(batch["question"], decoding_strategy="sampling")["predictions"]
        else:
            programs = batch["program"]
        if not self.nmn.training:  # (Module.train() walks every submodule: 0.5 ms per step)
            self.nmn.train()
        self.nmn.report_batch_metrics = self.report_metrics
        out = self.nmn(batch["image"], programs, batch["answer"])
        loss = out["loss"].mean()
        # data parallel: the reference's loss is the mean over the whole batch (module_training_trainer.py:91);
        # weighting the local mean by n_local * world / n_global keeps that exact for unequal shards
        if self._early is not None:
            self._early.arm()
        (loss * parallel.mean_weight(out["loss"].numel(), out["loss"].device)).backward()
        parallel.all_reduce_gradients(self.optimizer.arenas, self.optimizer.loose, early=self._early)
        self.optimizer.step()
        self.iteration += 1
        return {"loss": loss.detach(), "metrics": out.get("metrics")}


class ProgramPriorStep(StepBase):
    """One program-prior iteration (reference: probnmn/trainers/program_prior_trainer.py:79-90 and
    _trainer.py:135-151): mean over the batch of the per-sequence cross entropy of the LSTM language model,
    backward, clamp to [-5, 5], Adam.  ``after_validation`` takes 1 / perplexity (higher is better), as
    program_prior_trainer.py:112 does."""

    def __init__(self, program_prior, lr: float = 1e-2, weight_decay: float = 0.0, lr_gamma: float = 0.5,
                 lr_patience: int = 3):
        self.prior = program_prior
        self.optimizer = ClampAdam(program_prior.parameters(), lr=lr, weight_decay=weight_decay, clamp=5.0)
        self.models = {"program_prior": program_prior}
        self._init_schedule(lr_gamma, lr_patience)
        self.iteration = 0

    def step(self, batch: Dict[str, torch.Tensor]) -> Dict[str, Any]:
        self.optimizer.zero_grad()
        if not self.prior.training:
            self.prior.train()
        loss_rows = self.prior(batch["program"], need_predictions=False)["loss"]
        loss = loss_rows.mean()
        (loss * parallel.mean_weight(loss_rows.numel(), loss_rows.device)).backward()
        parallel.all_reduce_gradients([], self.optimizer.loose)
        self.optimizer.step()
        self.iteration += 1
        return {"loss": loss.detach()}
