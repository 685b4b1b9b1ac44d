"""What the three iterations share with the reference's ``_Trainer`` besides the step itself
(reference: probnmn/trainers/_trainer.py:103-130, 208-270): ONE Adam over every trainable model, a
``ReduceLROnPlateau(mode="max", factor=LR_GAMMA, patience=LR_PATIENCE, threshold=1e-3)`` stepped on the
validation metric, and a checkpoint whose layout is the reference ``CheckpointManager``'s --
``{<model name>: state_dict, ..., "optimizer": ..., "scheduler": ..., "iteration": n}``
(probnmn/utils/checkpointing.py:68-105,113-157) -- so that a file written by either side loads into the
other (the optimizer entry is ``torch.optim.Adam``'s state layout, see probnmn.optim.ClampAdam)."""
from typing import Any, Dict

import torch

from probnmn import parallel


class StepBase:
    #: name -> model, with the reference trainers' keys ("program_generator", "question_reconstructor", "nmn")
    models: Dict[str, torch.nn.Module]

    def _init_schedule(self, lr_gamma: float, lr_patience: int) -> None:
        self.lr_scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(
            self.optimizer, mode="max", factor=lr_gamma, patience=lr_patience, threshold=1e-3)

    def after_validation(self, metric: float) -> float:
        """``_Trainer.after_validation`` minus logging / file writing: lr scheduling on the validation
        metric (higher is better).  Returns the learning rate the next step will use.  Under data
        parallelism every rank must call it with the same (all-reduced) metric."""
        self.lr_scheduler.step(metric)
        return self.optimizer.param_groups[0]["lr"]

    def close(self) -> None:
        """Detach this trainer's gradient hooks from the models (another trainer may hook the same
        parameters afterwards -- the reference's phase pipeline reuses the NMN across trainers)."""
        for m in getattr(self, "models", {}).values():
            engine = getattr(m, "engine", None)
            if engine is not None:  # (a joint step's CU budgets must not outlive it: the next trainer may have the chip alone)
                engine.conv_cus = engine.wgrad_cus = 0
        early = getattr(self, "_early", None)
        if early is not None:
            early.remove()
            for e in getattr(early, "engines", ()):
                # (only if the callback is still this reducer's: a trainer built later over the same NMN owns it now)
                if getattr(e, "_grad_piece_owner", None) is early:
                    e.on_grad_piece = None
                    e._grad_piece_owner = None
            self._early = None

    # ---- checkpoint (reference layout) -----------------------------------------------------------
    def state_dict(self) -> Dict[str, Any]:
        out: Dict[str, Any] = {name: m.state_dict() for name, m in self.models.items()}
        out["optimizer"] = self.optimizer.state_dict()
        out["scheduler"] = self.lr_scheduler.state_dict()
        out["iteration"] = self.iteration
        return out

    def load_state_dict(self, checkpoint: Dict[str, Any]) -> int:
        """Loads whatever of {models, "optimizer", "scheduler"} the checkpoint holds (missing entries are
        skipped, as CheckpointManager.load does) and returns its iteration (-1 if absent)."""
        for name, m in self.models.items():
            if name in checkpoint:
                m.load_state_dict(checkpoint[name])
        if "optimizer" in checkpoint:
            self.optimizer.load_state_dict(checkpoint["optimizer"])
        if "scheduler" in checkpoint:
            self.lr_scheduler.load_state_dict(checkpoint["scheduler"])
        it = int(checkpoint.get("iteration", -1))
        if it >= 0:
            self.iteration = it
        return it
