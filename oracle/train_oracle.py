"""CPU restatement of one module-training iteration (test infrastructure; also the timed
``cpu_baseline`` of bench.py).  Follows reference probnmn/trainers/_trainer.py:135-151,193
(zero_grad -> batch -> _do_iteration -> optimizer.step) and
module_training_trainer.py:88-98 (NMN forward on the programs, mean loss, backward, element-wise
clamp of every gradient to [-5, 5]); the optimizer is the reference's own ``torch.optim.Adam``
(_trainer.py:103-108)."""
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
        self.optimizer.zero_grad()
        out = nmn_oracle.nmn_forward(self.params, self.index_to_token, batch["image"], batch["program"],
                                     batch["answer"])
        loss = out["loss"].mean()
        loss.backward()
        for p in self.params.values():
            if p.grad is not None:
                p.grad.clamp_(min=-5, max=5)
        self.optimizer.step()
        return {"loss": loss.detach(), "predictions": out["predictions"], "valid": out["valid"]}
