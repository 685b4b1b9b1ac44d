"""CPU restatement of the REINFORCE / ELBO arithmetic (test infrastructure -- see oracle/__init__).
Follows reference probnmn/modules/elbo.py:12-34 (Reinforce), :61-89 (_ElboWithReinforce._forward),
:130-161 (QuestionCodingElbo.forward), :220-280 (JointTrainingElbo.forward).  Pinned against the
real reference's outputs in tests/golden/elbo_known.json (oracle/make_golden.py)."""
from typing import Dict

import torch


class Reinforce:
    def __init__(self, baseline_decay: float = 0.99):
        self.baseline = 0.0
        self.decay = baseline_decay

    def __call__(self, inputs: torch.Tensor, reward: torch.Tensor) -> torch.Tensor:
        centered = reward.detach() - self.baseline
        self.baseline += self.decay * centered.mean().item()  # NOT an EMA (elbo.py:33)
        return inputs * centered


def elbo_forward(reinforce: Reinforce, beta: float, logq, logp_rec, reward) -> Dict[str, torch.Tensor]:
    kl = reinforce(logq, reward) - beta * logq
    elbo = logp_rec - kl
    return {"reconstruction_likelihood": logp_rec.mean(), "kl_divergence": kl.mean(), "elbo": elbo.mean(),
            "reinforce_reward": reward.mean()}


def question_coding_elbo(reinforce, beta, pg_loss, qr_loss, prior_loss):
    logq, rec, prior = -pg_loss, -qr_loss, -prior_loss
    reward = rec + beta * (prior - logq)
    return elbo_forward(reinforce, beta, logq, rec, reward)


def joint_training_elbo(reinforce, beta, gamma, objective, pg_loss, qr_loss, prior_loss, nmn_loss):
    if objective == "baseline":
        reward = -nmn_loss
        out = {"elbo": reinforce(pg_loss, reward).mean(), "reinforce_reward": reward.mean()}
    else:
        logq, rec, prior, ans = -pg_loss, -qr_loss, -prior_loss, -nmn_loss
        reward = rec + beta * prior - beta * logq + gamma * ans
        out = elbo_forward(reinforce, beta, logq, rec, reward)
    out["nmn_loss"] = nmn_loss.mean()
    return out
