"""CPU restatement of one module-training iteration (test infrastructure; also the timed
``cpu_baseline`` of bench.py).  Follows reference probnmn/trainers/_trainer.py:135-151,193
(zero_grad -> batch -> _do_iteration -> optimizer.step) and
module_training_trainer.py:88-98 (NMN forward on the programs, mean loss, backward, element-wise
clamp of every gradient to [-5, 5]); the optimizer is the reference's own ``torch.optim.Adam``
(_trainer.py:103-108).

``zero_grad`` is restated as torch 1.4.0 runs it (the reference pins torch==1.4.0, requirements.txt:6): gradients that
exist are ZEROED IN PLACE, not dropped -- so a parameter that has received a gradient once keeps being updated by Adam
(momentum, per-parameter step count) in iterations whose batch does not use its module.  torch >= 2.0 defaults to
``set_to_none=True``, under which Adam skips such a parameter: a different trajectory as soon as a batch leaves a
module out (tests/test_trajectory_gpu.py)."""
from typing import Dict

import torch

from oracle import nmn_oracle


class OracleModuleTrainer:
    def __init__(self, state_dict: Dict[str, torch.Tensor], index_to_token, lr: float = 1e-4,
                 weight_decay: float = 0.0):
        self.params = {k: v.detach().clone().contiguous().requires_grad_(True) for k, v in state_dict.items()}
        self.index_to_token = index_to_token
        self.optimizer = torch.optim.Adam(list(self.params.values()), lr=lr, weight_decay=weight_decay)

    def step(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        self.optimizer.zero_grad(set_to_none=False)  # (torch 1.4.0 semantics, see the module docstring)
        out = nmn_oracle.nmn_forward(self.params, self.index_to_token, batch["image"], batch["program"],
                                     batch["answer"])
        loss = out["loss"].mean()
        loss.backward()
        for p in self.params.values():
            if p.grad is not None:
                p.grad.clamp_(min=-5, max=5)
        self.optimizer.step()
        return {"loss": loss.detach(), "predictions": out["predictions"], "valid": out["valid"]}


class OracleJointTrainer:
    """CPU restatement of one joint-training iteration (reference
    probnmn/trainers/joint_training_trainer.py:128-198 + modules/elbo.py:220-280 +
    _trainer.py:103-108,135-151,193).  ``forced_programs`` replaces the ProgramGenerator's
    multinomial draw with the tokens another run sampled (the device's stream differs)."""

    def __init__(self, pg_sd, qr_sd, prior_sd, nmn_sd, index_to_token, objective="ours", alpha=100.0, beta=0.1,
                 gamma=1.0, delta=0.99, lr=1e-6, pg_steps=26, qr_steps=45):
        from oracle import elbo_oracle

        def leaf(sd):
            return {k: v.detach().clone().contiguous().requires_grad_(True) for k, v in sd.items()}

        self.pg, self.qr, self.nmn = leaf(pg_sd), leaf(qr_sd), leaf(nmn_sd)
        self.prior = {k: v.detach().clone() for k, v in prior_sd.items()}
        self.index_to_token = index_to_token
        self.objective, self.alpha, self.beta, self.gamma = objective, alpha, beta, gamma
        self.reinforce = elbo_oracle.Reinforce(delta)
        self.pg_steps, self.qr_steps = pg_steps, qr_steps
        params = list(self.pg.values()) + list(self.qr.values()) + list(self.nmn.values())
        self.optimizer = torch.optim.Adam(params, lr=lr)

    def step(self, batch, forced_programs=None):
        from oracle import elbo_oracle, seq2seq_oracle as so

        self.optimizer.zero_grad(set_to_none=False)  # (torch 1.4.0 semantics, see the module docstring)
        sup = batch["supervision"].nonzero().flatten()
        nosup = (1 - batch["supervision"]).nonzero().flatten()
        q, img, ans = batch["question"][nosup], batch["image"][nosup], batch["answer"][nosup]
        pg_out = so.seq2seq_forward(self.pg, q, None, "sampling", self.pg_steps, forced_predictions=forced_programs)
        z = pg_out["predictions"]
        qr_out = so.seq2seq_forward(self.qr, z, q, "sampling", self.qr_steps)
        nmn_out = nmn_oracle.nmn_forward(self.nmn, self.index_to_token, img, z, ans)
        with torch.no_grad():
            prior_loss = so.program_prior_loss(self.prior, z)
        out = elbo_oracle.joint_training_elbo(self.reinforce, self.beta, self.gamma, self.objective, pg_out["loss"],
                                              qr_out["loss"], prior_loss, nmn_out["loss"])
        nmn_loss = out.pop("nmn_loss")
        loss = self.gamma * nmn_loss - out["elbo"]
        if self.objective == "ours":
            prog, ques = batch["program"][sup], batch["question"][sup]
            pg_sup = so.seq2seq_forward(self.pg, ques, prog, "sampling")["loss"].mean()
            qr_sup = so.seq2seq_forward(self.qr, prog, ques, "sampling")["loss"].mean()
            loss = loss + self.alpha * (pg_sup + qr_sup)
        loss.backward()
        for p in list(self.pg.values()) + list(self.qr.values()) + list(self.nmn.values()):
            if p.grad is not None:
                p.grad.clamp_(min=-5, max=5)
        grads = {"pg": {k: (None if v.grad is None else v.grad.clone()) for k, v in self.pg.items()},
                 "qr": {k: (None if v.grad is None else v.grad.clone()) for k, v in self.qr.items()},
                 "nmn": {k: (None if v.grad is None else v.grad.clone()) for k, v in self.nmn.items()}}
        self.optimizer.step()
        return {"objective": loss.detach(), "nmn_loss": nmn_loss.detach(), "elbo": {k: v.detach() for k, v in out.items()},
                "programs": z, "grads": grads, "baseline": self.reinforce.baseline}


class OracleQuestionCodingTrainer:
    """CPU restatement of one question-coding iteration (reference
    probnmn/trainers/question_coding_trainer.py:109-168 + modules/elbo.py:130-161)."""

    def __init__(self, pg_sd, qr_sd, prior_sd, objective="ours", alpha=100.0, beta=0.1, delta=0.99, lr=1e-3,
                 pg_steps=26):
        from oracle import elbo_oracle

        def leaf(sd):
            return {k: v.detach().clone().contiguous().requires_grad_(True) for k, v in sd.items()}

        self.pg, self.qr = leaf(pg_sd), leaf(qr_sd)
        self.prior = {k: v.detach().clone() for k, v in prior_sd.items()}
        self.objective, self.alpha, self.beta = objective, alpha, beta
        self.reinforce = elbo_oracle.Reinforce(delta)
        self.pg_steps = pg_steps
        self.optimizer = torch.optim.Adam(list(self.pg.values()) + list(self.qr.values()), lr=lr)

    def step(self, batch, forced_programs=None):
        from oracle import elbo_oracle, seq2seq_oracle as so

        self.optimizer.zero_grad(set_to_none=False)  # (torch 1.4.0 semantics, see the module docstring)
        sup = batch["supervision"].nonzero().flatten()
        nosup = (1 - batch["supervision"]).nonzero().flatten()
        prog, ques = batch["program"][sup], batch["question"][sup]
        pg_sup = so.seq2seq_forward(self.pg, ques, prog, "sampling")["loss"].mean()
        qr_sup = so.seq2seq_forward(self.qr, prog, ques, "sampling")["loss"].mean()
        out = {}
        if self.objective == "baseline":
            loss = pg_sup + qr_sup
        else:
            q = batch["question"][nosup]
            pg_out = so.seq2seq_forward(self.pg, q, None, "sampling", self.pg_steps, forced_predictions=forced_programs)
            z = pg_out["predictions"]
            qr_out = so.seq2seq_forward(self.qr, z, q, "sampling")
            with torch.no_grad():
                prior_loss = so.program_prior_loss(self.prior, z)
            out = elbo_oracle.question_coding_elbo(self.reinforce, self.beta, pg_out["loss"], qr_out["loss"], prior_loss)
            loss = -out["elbo"] + self.alpha * pg_sup + self.alpha * qr_sup
        loss.backward()
        for p in list(self.pg.values()) + list(self.qr.values()):
            if p.grad is not None:
                p.grad.clamp_(min=-5, max=5)
        grads = {"pg": {k: (None if v.grad is None else v.grad.clone()) for k, v in self.pg.items()},
                 "qr": {k: (None if v.grad is None else v.grad.clone()) for k, v in self.qr.items()}}
        self.optimizer.step()
        return {"objective": loss.detach(), "elbo": {k: v.detach() for k, v in out.items()}, "grads": grads,
                "baseline": self.reinforce.baseline}
