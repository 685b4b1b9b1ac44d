"""REINFORCE estimator and the two evidence lower bounds -- class surface of the reference's
``probnmn.modules.elbo`` (reference: probnmn/modules/elbo.py:12-280).

Arithmetic is the reference's, including what looks odd: the "moving average" baseline is
``b += decay * mean(R - b)`` (not an EMA), the surrogate is ``logq * (R - b) - beta * logq``.
Two things differ in mechanics: the baseline lives on the device (the reference's
``.mean().item()`` costs a host sync per step), and under data parallelism the batch mean that
updates it is the GLOBAL mean (sum and count are all-reduced), so every rank keeps the same
baseline -- SURVEY.md 8(e).
"""
from typing import Dict

import torch
from torch import nn

from probnmn import parallel


class Reinforce(nn.Module):
    def __init__(self, baseline_decay: float = 0.99):
        super().__init__()
        self._baseline = None  # 0-dim tensor on the reward's device; not part of state_dict (as in the reference)
        self._baseline_decay = baseline_decay

    @property
    def _reinforce_baseline(self) -> float:
        return 0.0 if self._baseline is None else float(self._baseline)

    def forward(self, inputs: torch.Tensor, reward: torch.Tensor) -> torch.Tensor:
        reward = reward.detach()
        if self._baseline is None or self._baseline.device != reward.device:
            value = 0.0 if self._baseline is None else float(self._baseline)
            self._baseline = torch.full((), value, dtype=reward.dtype, device=reward.device)
        centered = reward - self._baseline
        stats = torch.stack((centered.sum(), centered.new_tensor(float(centered.numel()))))
        stats = parallel.all_reduce_scalars(stats)
        self._baseline = self._baseline + self._baseline_decay * stats[0] / stats[1]
        return inputs * centered


class _ElboWithReinforce(nn.Module):
    def __init__(self, beta: float = 0.1, baseline_decay: float = 0.99):
        super().__init__()
        self._reinforce = Reinforce(baseline_decay=baseline_decay)
        self._beta = beta

    def _forward(self, inference_likelihood, reconstruction_likelihood, reinforce_reward) -> Dict[str, torch.Tensor]:
        kl_divergence = self._reinforce(inference_likelihood, reinforce_reward) - self._beta * inference_likelihood
        fully_monte_carlo_elbo = reconstruction_likelihood - kl_divergence
        return {
            "reconstruction_likelihood": reconstruction_likelihood.mean(),
            "kl_divergence": kl_divergence.mean(),
            "elbo": fully_monte_carlo_elbo.mean(),
            "reinforce_reward": reinforce_reward.mean(),
        }


class QuestionCodingElbo(_ElboWithReinforce):
    def __init__(self, program_generator, question_reconstructor, program_prior, beta: float = 0.1,
                 baseline_decay: float = 0.99):
        super().__init__(beta, baseline_decay)
        self._program_generator = program_generator
        self._question_reconstructor = question_reconstructor
        self._program_prior = program_prior

    def forward(self, question_tokens: torch.LongTensor):
        pg_out = self._program_generator(question_tokens, decoding_strategy="sampling")
        sampled_programs = pg_out["predictions"]
        qr_out = self._question_reconstructor(sampled_programs, question_tokens, decoding_strategy="sampling")
        logprobs_reconstruction = -qr_out["loss"]
        logprobs_generation = -pg_out["loss"]
        with torch.no_grad():  # frozen model whose output only enters the detached reward
            logprobs_prior = -self._program_prior(sampled_programs)["loss"]
        reinforce_reward = logprobs_reconstruction + self._beta * (logprobs_prior - logprobs_generation)
        return super()._forward(logprobs_generation, logprobs_reconstruction, reinforce_reward)


class JointTrainingElbo(_ElboWithReinforce):
    def __init__(self, program_generator, question_reconstructor, program_prior, nmn, beta: float = 0.1,
                 gamma: float = 10, baseline_decay: float = 0.99, objective: str = "ours"):
        super().__init__(beta, baseline_decay)
        self._program_generator = program_generator
        self._question_reconstructor = question_reconstructor
        self._program_prior = program_prior
        self._nmn = nmn
        self._gamma = gamma
        self._objective = objective

    def forward(self, question_tokens, image_features, answer_tokens):
        pg_out = self._program_generator(question_tokens, decoding_strategy="sampling")
        sampled_programs = pg_out["predictions"]
        qr_out = self._question_reconstructor(sampled_programs, question_tokens, decoding_strategy="sampling")
        nmn_out = self._nmn(image_features, sampled_programs, answer_tokens)
        if self._objective == "baseline":
            reinforce_reward = -nmn_out["loss"]
            output_dict = {
                "elbo": self._reinforce(pg_out["loss"], reinforce_reward).mean(),
                "reinforce_reward": reinforce_reward.mean(),
            }
        else:
            logprobs_reconstruction = -qr_out["loss"]
            logprobs_generation = -pg_out["loss"]
            with torch.no_grad():
                logprobs_prior = -self._program_prior(sampled_programs)["loss"]
            logprobs_answering = -nmn_out["loss"]
            reinforce_reward = (logprobs_reconstruction + self._beta * logprobs_prior
                                - self._beta * logprobs_generation + self._gamma * logprobs_answering)
            output_dict = super()._forward(logprobs_generation, logprobs_reconstruction, reinforce_reward)
        output_dict["nmn_loss"] = nmn_out["loss"].mean()
        return output_dict
