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
        stats = torch.stack((centered.sum(), torch.full_like(self._baseline, float(centered.numel()))))
        stats = parallel.all_reduce_scalars(stats)
        # (a rank whose shard holds no sampled rows still takes part in the collective -- see `idle`;
        # with no rows anywhere the baseline stays where it is)
        self._baseline = self._baseline + self._baseline_decay * stats[0] / stats[1].clamp(min=1.0)
        return inputs * centered

    def idle(self, device, dtype=torch.float32) -> None:
        """Data parallel: this rank has no sampled rows in this iteration.  Takes part in the baseline's
        all-reduce with an empty contribution, so that every rank issues the same collectives and ends
        the iteration with the same baseline."""
        empty = torch.zeros(0, dtype=dtype, device=device)
        self.forward(empty, empty)


class _FusedElbo(torch.autograd.Function):
    """``pnmn_elbo_rows``: reward, centring, KL surrogate, ELBO and the five batch sums in one launch; the
    backward is analytic (d sum(elbo) / d pg_loss[n] = (R[n] - b) - beta, d / d qr_loss[n] = -1, the reward
    itself is a constant of the estimator -- reference elbo.py:28-34,76-82)."""

    @staticmethod
    def forward(ctx, pg_loss, qr_loss, prior_loss, nmn_loss, baseline, beta, gamma):
        n = pg_loss.numel()
        dev = pg_loss.device
        # (zeros, not empty: with no sampled rows the kernel returns without writing, and the means and the
        # baseline update below must then see zeros, not whatever the allocator handed out)
        sums = torch.zeros(6, dtype=torch.float32, device=dev)
        dpg = torch.empty(n, dtype=torch.float32, device=dev)
        from probnmn import _hip

        keep = [t.detach().contiguous() if t is not None else None for t in (pg_loss, qr_loss, prior_loss, nmn_loss)]
        _hip.check(_hip.lib().pnmn_elbo_rows(*[0 if t is None else t.data_ptr() for t in keep], baseline.data_ptr(),
                                             float(beta), float(gamma), n, sums.data_ptr(), dpg.data_ptr(),
                                             _hip.stream_ptr(dev)), "elbo_rows")
        ctx.save_for_backward(dpg)
        ctx.has_nmn = nmn_loss is not None
        return sums

    @staticmethod
    def backward(ctx, dsums):
        (dpg,) = ctx.saved_tensors
        # sums = [sum rec, sum kl, sum elbo, sum R, sum nmn, sum c]; with rec = -qr, kl = -pg c + beta pg, elbo = rec - kl:
        #   d/d pg[n] = -dsums[1] (c - beta) + dsums[2] (c - beta);  d/d qr[n] = -dsums[0] - dsums[2];  d/d nmn[n] = dsums[4]
        d_pg = (dsums[2] - dsums[1]) * dpg
        d_qr = -(dsums[0] + dsums[2]).expand_as(dpg)
        d_nmn = dsums[4].expand_as(dpg) if ctx.has_nmn else None
        return d_pg, d_qr, None, d_nmn, None, None, None


class _Objective(torch.autograd.Function):
    """``pnmn_joint_objective``: the ELBO combination over the sampled rows, the supervised rows' mean cross entropies,
    the objective J = w_u (gamma mean(nmn) - mean(elbo)) + w_s alpha (mean(pg_sup) + mean(qr_sup)) and every per-row
    derivative in one launch; backward is ONE multiply of the stored derivatives by the incoming dJ.  Replaces the
    ~30 scalar torch ops (and their autograd nodes) between the models' per-row losses and ``backward()`` of a
    question-coding / joint-training iteration (reference question_coding_trainer.py:128-165,
    joint_training_trainer.py:150-191) -- at 128 questions per GPU that chain alone was 0.5 ms of host time."""

    @staticmethod
    def forward(ctx, pg, qr, prior, nmn, pg_sup, baseline, w_u, w_s, alpha, beta, gamma, decay, update_baseline, n, m):
        from probnmn import _hip

        dev = baseline.device
        keep = [None if t is None else t.detach().contiguous() for t in (pg, qr, prior, nmn, pg_sup, w_u, w_s)]
        out = torch.zeros(11, dtype=torch.float32, device=dev)  # stats[10] + J
        has_nmn = nmn is not None
        sizes = (n, n + m, n if has_nmn else 0, m)
        grads = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        ptrs, at = [], 0
        for k in sizes:
            ptrs.append(grads.data_ptr() + 4 * at if k else 0)
            at += k
        p = [0 if t is None else t.data_ptr() for t in keep]
        _hip.check(_hip.lib().pnmn_joint_objective(
            p[0], p[1], p[2], p[3], p[4], baseline.data_ptr(), p[5], p[6], float(alpha), float(beta), float(gamma),
            float(decay), int(update_baseline), n, m, out.data_ptr(), out.data_ptr() + 40, ptrs[0], ptrs[1], ptrs[2],
            ptrs[3], _hip.stream_ptr(dev)), "joint_objective")
        ctx.save_for_backward(grads)
        ctx.sizes = sizes
        ctx.present = (pg is not None, qr is not None, has_nmn, pg_sup is not None)
        stats, objective = out[:10], out[10]
        ctx.mark_non_differentiable(stats)
        return objective, stats

    @staticmethod
    def backward(ctx, d_objective, _):
        (grads,) = ctx.saved_tensors
        grads = grads * d_objective
        d_pg, d_qr, d_nmn, d_pgs = grads.split(ctx.sizes)
        has_pg, has_qr, has_nmn, has_pgs = ctx.present
        return (d_pg if has_pg else None, d_qr if has_qr else None, None, d_nmn if has_nmn else None,
                d_pgs if has_pgs else None, None, None, None, None, None, None, None, None, None, None)


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

    def objective(self, generation_loss, reconstruction_rows, prior_loss, nmn_loss, supervised_generation_loss,
                  w_unsup, w_sup, alpha: float, gamma: float, n: int, m: int):
        """The "ours" objective of an iteration from the models' per-row losses, on the device in one launch (see
        ``_Objective``).  ``reconstruction_rows``: the n sampled rows' reconstruction losses followed by the m supervised
        ones.  Returns (J, dict of the detached batch statistics the trainers report).  The REINFORCE baseline is
        updated here (in the kernel in a single process; from the all-reduced sum under data parallelism)."""
        r = self._reinforce
        dev = reconstruction_rows.device
        if r._baseline is None or r._baseline.device != dev:
            value = 0.0 if r._baseline is None else float(r._baseline)
            r._baseline = torch.full((), value, dtype=torch.float32, device=dev)
        single = parallel.world() == 1
        w_u = w_unsup if isinstance(w_unsup, torch.Tensor) else None
        w_s = w_sup if isinstance(w_sup, torch.Tensor) else None
        J, stats = _Objective.apply(generation_loss, reconstruction_rows, prior_loss, nmn_loss, supervised_generation_loss,
                                    r._baseline, w_u, w_s, alpha, self._beta, gamma, r._baseline_decay, single, n, m)
        if not single:
            with torch.no_grad():
                total = parallel.all_reduce_scalars(torch.stack((stats[5], stats[9])))
                r._baseline = r._baseline + r._baseline_decay * total[0] / total[1].clamp(min=1.0)
        out = {"reconstruction_likelihood": stats[0], "kl_divergence": stats[1], "elbo": stats[2], "reinforce_reward": stats[3],
               "nmn_loss": stats[4], "program_generation_gt": stats[6], "question_reconstruction_gt": stats[7]}
        return J, out

    def _fused(self, generation_loss, reconstruction_loss, prior_loss, nmn_loss, gamma: float) -> Dict[str, torch.Tensor]:
        """The "ours" objectives on the device in one launch (same arithmetic as ``combine`` -> ``_forward``
        -> ``Reinforce.forward``; the baseline update and its data-parallel all-reduce stay here)."""
        r = self._reinforce
        if r._baseline is None or r._baseline.device != generation_loss.device:
            value = 0.0 if r._baseline is None else float(r._baseline)
            r._baseline = torch.full((), value, dtype=torch.float32, device=generation_loss.device)
        n = generation_loss.numel()
        sums = _FusedElbo.apply(generation_loss, reconstruction_loss, prior_loss, nmn_loss, r._baseline, self._beta, gamma)
        with torch.no_grad():
            stats = torch.stack((sums[5], torch.full_like(sums[5], float(n))))
            stats = parallel.all_reduce_scalars(stats)
            r._baseline = r._baseline + r._baseline_decay * stats[0] / stats[1].clamp(min=1.0)
        means = sums / max(n, 1)
        out = {"reconstruction_likelihood": means[0], "kl_divergence": means[1], "elbo": means[2],
               "reinforce_reward": means[3]}
        if nmn_loss is not None:
            out["nmn_loss"] = means[4]
        return out


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
        with torch.no_grad():  # frozen model whose output only enters the detached reward
            prior_out = self._program_prior(sampled_programs)
        qr_out = self._question_reconstructor(sampled_programs, question_tokens, decoding_strategy="sampling")
        return self.combine(pg_out["loss"], qr_out["loss"], prior_out["loss"])

    def combine(self, generation_loss, reconstruction_loss, prior_loss) -> Dict[str, torch.Tensor]:
        """The objective from the three per-example negative log-likelihoods of the SAMPLED programs
        (reference elbo.py:130-161); trainers that batch the model passes themselves call this."""
        if generation_loss.is_cuda:
            return self._fused(generation_loss, reconstruction_loss, prior_loss, None, 0.0)
        logprobs_reconstruction = -reconstruction_loss
        logprobs_generation = -generation_loss
        logprobs_prior = -prior_loss
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
        # (one stream: the recurrent kernels and the library GEMMs of these passes must not share the chip
        # with each other -- DESIGN.md 6; the batched joint step in probnmn.trainers overlaps the NMN trunk)
        qr_loss = self._question_reconstructor(sampled_programs, question_tokens, decoding_strategy="sampling")["loss"]
        prior_loss = None
        if self._objective != "baseline":
            with torch.no_grad():  # frozen model whose output only enters the detached reward
                prior_loss = self._program_prior(sampled_programs)["loss"]
        nmn_out = self._nmn(image_features, sampled_programs, answer_tokens)
        return self.combine(pg_out["loss"], qr_loss, prior_loss, nmn_out)

    def combine(self, generation_loss, reconstruction_loss, prior_loss, nmn_out) -> Dict[str, torch.Tensor]:
        """The objective from the per-example losses of the SAMPLED programs (reference
        elbo.py:220-280); trainers that batch the model passes themselves call this."""
        if self._objective == "baseline":
            reinforce_reward = -nmn_out["loss"]
            output_dict = {
                "elbo": self._reinforce(generation_loss, reinforce_reward).mean(),
                "reinforce_reward": reinforce_reward.mean(),
            }
        elif generation_loss.is_cuda:
            return self._fused(generation_loss, reconstruction_loss, prior_loss, nmn_out["loss"], self._gamma)
        else:
            logprobs_reconstruction = -reconstruction_loss
            logprobs_generation = -generation_loss
            logprobs_prior = -prior_loss
            logprobs_answering = -nmn_out["loss"]
            reinforce_reward = (logprobs_reconstruction + self._beta * logprobs_prior
                                - self._beta * logprobs_generation + self._gamma * logprobs_answering)
            output_dict = super()._forward(logprobs_generation, logprobs_reconstruction, reinforce_reward)
        output_dict["nmn_loss"] = nmn_out["loss"].mean()
        return output_dict
