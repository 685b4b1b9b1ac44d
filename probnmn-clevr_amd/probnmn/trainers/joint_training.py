"""Question-coding and joint-training iterations on one GPU / one DP rank (reference:
probnmn/trainers/question_coding_trainer.py:109-168, joint_training_trainer.py:128-198,
_trainer.py:135-151): split the batch into supervised / unsupervised examples, -ELBO (+ gamma *
answer loss) on the unsupervised ones, alpha-weighted teacher-forced cross entropies on the
supervised ones, backward, [gradient all-reduce], element-wise clamp to [-5, 5], one Adam over all
trainable models.

Data parallelism: every loss term is a mean over a data-dependent subset, so a mean of local means
would weight shards wrongly; each local mean is rescaled by (n_local * world / n_global) before
backward, which makes the averaged all-reduced gradient equal the single-process gradient
(SURVEY.md 8e).
"""
from typing import Any, Dict

import torch

from probnmn import parallel
from probnmn.modules.elbo import BRANCHES, JointTrainingElbo, QuestionCodingElbo
from probnmn.optim import ClampAdam


def _split_supervision(supervision: torch.Tensor):
    """Index tensors of supervised / unsupervised examples.  The split sizes are data dependent, so
    the host must know them; a CPU ``supervision`` tensor (what the data loader yields) costs no
    device sync."""
    sup_host = supervision.detach().cpu()
    return sup_host.nonzero().flatten(), (1 - sup_host).nonzero().flatten()


def _dp_weight(n_local: int, device) -> float:
    """n_local * world / n_global (1.0 in a single process)."""
    if parallel.world() == 1:
        return 1.0
    t = parallel.all_reduce_scalars(torch.tensor([float(n_local)], device=device))
    total = float(t.item())
    return n_local * parallel.world() / total if total > 0 else 0.0


class _TrainerBase:
    def _make_optimizer(self, models, lr, weight_decay):
        arenas = []
        params = []
        for m in models:
            if hasattr(m, "engine"):
                arenas.append(m.engine.ensure_arena())
                m.engine.direct_grads = True
            params.extend(m.parameters())
        return ClampAdam(params, arenas=arenas, lr=lr, weight_decay=weight_decay, clamp=5.0)

    def _finish(self, loss: torch.Tensor) -> None:
        loss.backward()
        parallel.all_reduce_gradients(self.optimizer.arenas, self.optimizer.loose)
        self.optimizer.step()
        self.iteration += 1


class QuestionCodingStep(_TrainerBase):
    def __init__(self, program_generator, question_reconstructor, program_prior, objective: str = "ours",
                 alpha: float = 100.0, beta: float = 0.1, delta: float = 0.99, lr: float = 1e-3,
                 weight_decay: float = 0.0):
        if objective not in ("ours", "baseline"):
            raise ValueError("objective must be 'ours' or 'baseline'")
        self.pg, self.qr, self.prior = program_generator, question_reconstructor, program_prior
        self.objective, self.alpha = objective, alpha
        self.prior.eval()
        self.elbo = QuestionCodingElbo(self.pg, self.qr, self.prior, beta=beta, baseline_decay=delta)
        self.optimizer = self._make_optimizer([self.pg, self.qr], lr, weight_decay)
        self.iteration = 0

    def step(self, batch: Dict[str, torch.Tensor]) -> Dict[str, Any]:
        self.optimizer.zero_grad()
        self.pg.train()
        self.qr.train()
        dev = batch["question"].device
        sup, nosup = _split_supervision(batch["supervision"])
        sup_d, nosup_d = sup.to(dev, non_blocking=True), nosup.to(dev, non_blocking=True)
        out: Dict[str, Any] = {}
        loss = torch.zeros((), device=dev)
        sup_branches = None
        if sup.numel():
            # the two teacher-forced passes depend on nothing: side streams, beside the ELBO chain
            prog, ques = batch["program"].to(dev)[sup_d], batch["question"][sup_d]
            sup_branches = [
                BRANCHES.run("sup_pg", dev, lambda: self.pg(ques, prog, decoding_strategy="sampling")["loss"].mean(),
                             ques, prog),
                BRANCHES.run("sup_qr", dev, lambda: self.qr(prog, ques, decoding_strategy="sampling")["loss"].mean(),
                             ques, prog),
            ]
        elbo_out = None
        if self.objective == "ours" and nosup.numel():
            elbo_out = self.elbo(batch["question"][nosup_d])
        if sup_branches is not None:
            BRANCHES.join(dev, sup_branches)
            pg_loss, qr_loss = sup_branches[0][0], sup_branches[1][0]
            w = _dp_weight(sup.numel(), dev)
            if self.objective == "baseline":
                loss = loss + w * pg_loss + w * qr_loss
            else:
                loss = loss + w * self.alpha * (pg_loss + qr_loss)
            out["loss"] = {"program_generation_gt": pg_loss.detach(), "question_reconstruction_gt": qr_loss.detach()}
        if elbo_out is not None:
            loss = loss - _dp_weight(nosup.numel(), dev) * elbo_out["elbo"]
            out["elbo"] = {k: v.detach() for k, v in elbo_out.items()}
        self._finish(loss)
        out["objective"] = loss.detach()
        return out


class JointTrainingStep(_TrainerBase):
    def __init__(self, program_generator, question_reconstructor, program_prior, nmn, objective: str = "ours",
                 alpha: float = 100.0, beta: float = 0.1, gamma: float = 1.0, delta: float = 0.99,
                 lr: float = 1e-6, weight_decay: float = 0.0):
        if objective not in ("ours", "baseline"):
            raise ValueError("objective must be 'ours' or 'baseline'")
        self.pg, self.qr, self.prior, self.nmn = program_generator, question_reconstructor, program_prior, nmn
        self.objective, self.alpha, self.gamma = objective, alpha, gamma
        self.prior.eval()
        self.elbo = JointTrainingElbo(self.pg, self.qr, self.prior, self.nmn, beta=beta, gamma=gamma,
                                      baseline_decay=delta, objective=objective)
        self.optimizer = self._make_optimizer([self.pg, self.qr, self.nmn], lr, weight_decay)
        self.iteration = 0

    def step(self, batch: Dict[str, torch.Tensor]) -> Dict[str, Any]:
        self.optimizer.zero_grad()
        for m in (self.pg, self.qr, self.nmn):
            m.train()
        self.nmn.report_batch_metrics = False
        dev = batch["question"].device
        sup, nosup = _split_supervision(batch["supervision"])
        sup_d, nosup_d = sup.to(dev, non_blocking=True), nosup.to(dev, non_blocking=True)
        if nosup.numel() == 0:
            raise ValueError("joint training needs at least one example without program supervision in the batch")
        sup_branches = None
        if self.objective == "ours" and sup.numel():
            prog, ques = batch["program"].to(dev)[sup_d], batch["question"][sup_d]
            sup_branches = [
                BRANCHES.run("sup_pg", dev, lambda: self.pg(ques, prog, decoding_strategy="sampling")["loss"].mean(),
                             ques, prog),
                BRANCHES.run("sup_qr", dev, lambda: self.qr(prog, ques, decoding_strategy="sampling")["loss"].mean(),
                             ques, prog),
            ]
        elbo_out = self.elbo(batch["question"][nosup_d], batch["image"][nosup_d], batch["answer"][nosup_d])
        nmn_loss = elbo_out.pop("nmn_loss")
        w = _dp_weight(nosup.numel(), dev)
        loss = w * (self.gamma * nmn_loss - elbo_out["elbo"])
        out: Dict[str, Any] = {"loss": {"nmn": nmn_loss.detach()}, "elbo": {k: v.detach() for k, v in elbo_out.items()}}
        if sup_branches is not None:
            BRANCHES.join(dev, sup_branches)
            pg_loss, qr_loss = sup_branches[0][0], sup_branches[1][0]
            loss = loss + _dp_weight(sup.numel(), dev) * self.alpha * (pg_loss + qr_loss)
            out["loss"]["program_generation_gt"] = pg_loss.detach()
            out["loss"]["question_reconstruction_gt"] = qr_loss.detach()
        self._finish(loss)
        out["objective"] = loss.detach()
        return out
