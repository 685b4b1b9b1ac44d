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
import os
import time
from typing import Any, Dict

import torch

from probnmn import _hip, parallel
from probnmn.modules.elbo import JointTrainingElbo, QuestionCodingElbo
from probnmn.optim import ClampAdam
from ._base import StepBase


def _split_supervision(supervision: torch.Tensor):
    """Index tensors of supervised / unsupervised examples.  The split sizes are data dependent, so
    the host must know them; a CPU ``supervision`` tensor (what the data loader yields) costs no
    device sync."""
    sup_host = supervision.detach().cpu()
    return sup_host.nonzero().flatten(), (1 - sup_host).nonzero().flatten()


def _index_to_device(index: torch.Tensor, dev: torch.device) -> torch.Tensor:
    """A host index tensor on the device through the pinned staging ring (``.to(dev)`` of a pageable tensor is a
    staged copy the host waits for: ~20 us each, twice per step)."""
    if dev.type != "cuda" or index.numel() == 0:
        return index.to(dev)
    return _hip.to_device(index.numpy(), dev).view(torch.long)


_dp_weight = parallel.mean_weight  # (n_local * world / n_global, see probnmn.parallel)


def _cat_padded(a: torch.Tensor, b: torch.Tensor, pad: int = 0) -> torch.Tensor:
    """Row-concatenate two token matrices, right-padding the narrower one."""
    w = max(a.size(1), b.size(1))
    if a.size(1) < w:
        a = torch.nn.functional.pad(a, (0, w - a.size(1)), value=pad)
    if b.size(1) < w:
        b = torch.nn.functional.pad(b, (0, w - b.size(1)), value=pad)
    return torch.cat((a, b), 0)


_SHARED_STREAMS: Dict[Any, "torch.cuda.Stream"] = {}


def shared_stream(dev: torch.device, role: str, priority: int = 0) -> "torch.cuda.Stream":
    """ONE stream per (device, role) for the whole process, however many trainers or loaders are built.  HIP multiplexes
    a process's streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default), and two streams that land on one
    queue run their kernels in order: every further ``torch.cuda.Stream()`` -- a second trainer's side stream, a second
    loader's -- shifted which streams share a queue, and whichever multi-stream workload was built LATER in a process ran
    5-13 ms per step slower on the GPU with the host unchanged (rounds 3-4: "allocator history"; round 5: the 1024-question
    step fed from a resident store 28.6 ms alone, 32.4 ms behind a 28x28 side object with its own side stream, 28.6 ms
    again with GPU_MAX_HW_QUEUES=2 -- profiles/ab/round5_stream_queues.txt).  (A high-priority trunk stream was measured:
    no effect, profiles/ab/r04m_ab.txt.)"""
    dev = torch.device(dev)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = (dev.index, role)
    if key not in _SHARED_STREAMS:
        _SHARED_STREAMS[key] = torch.cuda.Stream(device=dev, priority=priority)
    return _SHARED_STREAMS[key]


def shared_conv_cus(questions: int, banded: bool) -> int:
    """CUs the trunk's conv launches are cut for while it runs beside the seq2seq passes (0 = all of them).  Up to 128
    questions of 14x14 maps: what the passes' multi-CU kernels leave free -- eight workgroups per 16-row tile, one per CU
    (224 at 64 questions, 192 at 128).  Beyond, and for 28x28 maps (four band units per item): the whole chip.  Measured:
    JointTrainingStep.__init__ / DESIGN 5-6."""
    return 256 - 8 * (-(-questions // 16)) if (questions <= 128 and not banded) else 0


def stem_waits_for_encoder(questions: int) -> int:
    """``stem_after_encode`` mode of a step of that many questions: 2 (issued behind the generator's encoder pass and
    waiting for it on the GPU) from 256 questions on, 0 (first thing in the step) below."""
    return 2 if questions >= 256 else 0


def trunk_before_prior(questions: int) -> bool:
    """Module programs launched between the reconstructor and the prior pass (below 512 questions) or after all seq2seq
    passes are issued (from 512 on)."""
    return questions < 512


def _image_rows(images, rows_host, rows_dev):
    """(features, device row index) of a row subset of a batch's images: a tensor is read THROUGH the index by the
    layout kernel; rows of a resident feature store are named by their host indices (no device index needed)."""
    from probnmn.data.feature_store import ResidentRows

    if isinstance(images, ResidentRows):
        return images.subset(rows_host), None
    return images, rows_dev


class _TrainerBase(StepBase):
    # the backward passes of the generator's two decodes and the reconstructor's in ONE launch (False: pair + single -- A/B
    # aid and the reference point of tests/test_joint_gpu.py)
    group_decoder_backward = True
    # the seq2seq passes of an iteration as a static launch plan (probnmn.runtime.seq_plan: one autograd node, no torch op in
    # the passes) where the batch has supervised AND unsupervised rows and the models have the shapes the plan is built for;
    # False, or anything else: the eager passes below (one autograd node per kernel -- the plan's reference in
    # tests/test_seq_plan_gpu.py)
    use_plan = os.environ.get("PNMN_SEQ_PLAN", "1") != "0"

    def _plan(self, dev, n: int, m: int, tq: int, tp: int):
        """The cached plan for this shape signature (None: not plannable -- remembered, so the check is paid once)."""
        from probnmn.runtime.seq_plan import PlanUnsupported, Seq2SeqPlan

        plans = self.__dict__.setdefault("_plans", {})
        key = (dev.index, n, m, tq, tp, self.pg._max_decoding_steps)
        plan = plans.get(key)
        if plan is not None and plan is not False and not plan.still_valid():
            plan = None
        if plan is None:
            try:
                plan = Seq2SeqPlan(self.pg, self.qr, self.prior, dev, n, m, tq, tp)
            except PlanUnsupported:
                plan = False
            if len(plans) > 8:  # (a run with many batch shapes: keep the workspaces of the recent ones only)
                plans.pop(next(iter(plans)))
            plans[key] = plan
        return plan or None

    def _planned_encoder_workgroups(self, batch, sup, nosup) -> int:
        """Workgroups of the generator's encoder launch of this batch's plan (0: no plan, or a launch per layer)."""
        dev = batch["question"].device
        if not (self.use_plan and sup.numel() and nosup.numel() and dev.type == "cuda" and batch["program"].device == dev):
            return 0
        plan = self._plan(dev, int(nosup.numel()), int(sup.numel()), int(batch["question"].size(1)), int(batch["program"].size(1)))
        return plan.pg_encoder_workgroups if plan is not None else 0

    def _planned_passes(self, plan, batch, sup_d, nosup_d, host_programs, after_sampling, before_prior, after_encode):
        """``_seq2seq_passes`` of a batch with supervised and unsupervised rows through the launch plan: the same passes in
        the same order on the same stream, the same callbacks at the same points."""
        out = {"n_nosup": plan.n, "n_sup": plan.m}
        plan.run_encoder(batch["question"], batch["program"], nosup_d, sup_d)
        if after_encode is not None:
            out["after_encode"] = after_encode()
        z = plan.run_decoders()
        # (z is the plan's buffer, rewritten by the next iteration: what the caller gets to keep is a copy)
        out["programs"] = z.clone()
        if host_programs:
            out["programs_host"] = self._host_copy(z)
        if after_sampling is not None:
            out["after_sampling"] = after_sampling()
        plan.run_generator_finish()
        plan.run_reconstructor()
        # The plan's autograd node is created HERE, before the trunk is launched: autograd runs later-created nodes first, so
        # backward issues the NMN head, then the trunk's backward (the iteration's longest chain, on its own stream), then this
        # node's launch list -- created after the trunk's node it held the trunk's backward up by the ~0.5 ms of host time its
        # own replay takes (profiles/r06d_b128_timeline.txt: trunk backward 1.6 ms behind the objective).
        loss_s, loss_t, loss_q = plan.losses()
        if before_prior is not None:
            out["before_prior"] = before_prior(out["programs_host"])
        out["prior"] = plan.run_prior()
        out["pg"] = {"loss": loss_s, "predictions": out["programs"]}
        out["pg_sup_rows"], out["qr_rows"] = loss_t, loss_q
        return out

    def _make_optimizer(self, models, lr, weight_decay):
        arenas = []
        params = []
        for m in models:
            if hasattr(m, "engine"):
                arenas.append(m.engine.ensure_arena())
                m.engine.direct_grads = True
            params.extend(m.parameters())
        optimizer = ClampAdam(params, arenas=arenas, lr=lr, weight_decay=weight_decay, clamp=5.0)
        # data parallel: the NMN's fully connected layer (205 MB of gradient) is the first thing backward
        # finishes -- its all-reduce starts from the gradient hook and runs beside the NMN trunk backward
        # (plain grouped launches), well before the multi-CU recurrent kernels of the seq2seq backward,
        # which want the whole chip to themselves
        big = [p for p in optimizer.loose if p.numel() >= (1 << 20)]
        self._early = parallel.early_reducer_for(big, [m.engine for m in models if hasattr(m, "engine")])
        return optimizer

    def _finish(self, loss: torch.Tensor) -> None:
        early = getattr(self, "_early", None)
        if early is not None:
            early.arm()
        if loss.requires_grad:  # (false only for a data-parallel shard without any row: it contributes zeros)
            _hip.mark("backward begins")
            loss.backward()
            _hip.mark("backward issued")
        side = getattr(self, "_side", None)
        engine = getattr(getattr(self, "nmn", None), "engine", None)
        # The NMN's share of the optimiser step on the NMN's stream (single process; JointTrainingStep.overlap_optimizer): 28
        # bytes per parameter over the 63 M of the trunk arena and the fully connected layers are 0.33 ms with nothing beside
        # them at the end of the iteration -- 5 % of the 128-question step.  Launched on the side stream behind both backward
        # passes, they run beside the next iteration's generator passes; whatever reads NMN parameters next waits for
        # engine.params_ready (the stem and the trunk are on that stream anyway, the head waits for it).
        overlap = (getattr(self, "overlap_optimizer", False) and side is not None and engine is not None and parallel.world() == 1
                   and loss.requires_grad)
        if side is not None and not overlap:  # the NMN's backward ran on its own stream: gradients are used below on this one
            torch.cuda.current_stream(side.device).wait_stream(side)
        parallel.all_reduce_gradients(self.optimizer.arenas, self.optimizer.loose, early=getattr(self, "_early", None))
        if overlap:
            side.wait_stream(torch.cuda.current_stream(side.device))  # (the FC layers' gradients come from this stream)
            self.optimizer.step(side_stream=side, side_params=self._nmn_loose_ids())
            ready = torch.cuda.Event()
            ready.record(side)
            engine.params_ready = ready
        else:
            self.optimizer.step()
        self.iteration += 1
        if engine is not None:
            # the step's CU budgets end with it (everything is queued): a validation pass or another trainer that runs
            # this network next has the chip to itself
            engine.conv_cus = engine.wgrad_cus = 0

    def _nmn_loose_ids(self):
        ids = self.__dict__.get("_nmn_loose")
        if ids is None:
            ids = self.__dict__["_nmn_loose"] = {id(p) for p in self.nmn.parameters()}
        return ids

    def settle(self) -> None:
        """Make the current stream wait for an optimiser step still running on the NMN's stream (``overlap_optimizer``):
        call before reading NMN parameters outside the models' own methods (checkpoints do, through ``state_dict``)."""
        engine = getattr(getattr(self, "nmn", None), "engine", None)
        ready = getattr(engine, "params_ready", None)
        if ready is not None:
            torch.cuda.current_stream().wait_event(ready)

    def state_dict(self):
        self.settle()
        return super().state_dict()

    def close(self) -> None:
        self.settle()
        super().close()

    def _host_copy(self, tokens: torch.Tensor):
        """Start the device -> host copy of the sampled programs into a (cached) pinned buffer and
        return (host tensor, event).  Queued right behind the sampling decode, the copy completes
        while the GPU works through the passes launched after it, so the host can compile and
        schedule the NMN launches for the samples without the GPU ever waiting for it."""
        cache = self.__dict__.setdefault("_pinned_programs", {})
        key = tuple(tokens.shape)
        if key not in cache:
            cache[key] = torch.empty(key, dtype=tokens.dtype, pin_memory=True)
        host = cache[key]
        host.copy_(tokens, non_blocking=True)
        event = torch.cuda.Event()
        event.record()
        return host, event

    def _seq2seq_passes(self, batch, sup_d, nosup_d, supervised: bool, sampled: bool, prior: bool,
                        reconstruct: bool = True, host_programs: bool = False, after_sampling=None,
                        before_prior=None, after_encode=None):
        """All ProgramGenerator / QuestionReconstructor / ProgramPrior passes of one iteration, with the
        rows of the reference's separate calls batched into as few recurrent launches as the data
        dependencies allow (the persistent LSTM / decoder kernels are latency bound: a launch over
        more rows costs the same time):

          * one ProgramGenerator encoder pass over every question it will decode, then the
            teacher-forced decode of the supervised rows and the sampling decode of the others;
          * ONE QuestionReconstructor pass: both of the reference's calls are teacher-forced on the
            question, with the sampled programs resp. the ground-truth programs as source.

        Rows are independent in every model, so each row's loss equals the reference's separate
        calls' (question_coding_trainer.py:128-160, joint_training_trainer.py:150-190).
        ``reconstruct=False`` skips the reconstruction of the samples where the objective does not
        use it (the reference's joint "baseline" objective evaluates and discards it, elbo.py:236-251)."""
        dev = batch["question"].device
        out = {}
        n_sup, n_nosup = (sup_d.numel() if supervised else 0), (nosup_d.numel() if sampled else 0)
        if n_sup == 0 and n_nosup == 0:
            out["n_nosup"], out["n_sup"] = 0, 0
            return out
        question = batch["question"]
        if (self.use_plan and n_sup and n_nosup and prior and reconstruct and dev.type == "cuda" and torch.is_grad_enabled()
                and self.pg.training and self.qr.training and batch["program"].device == dev):
            plan = self._plan(dev, n_nosup, n_sup, int(question.size(1)), int(batch["program"].size(1)))
            if plan is not None:
                return self._planned_passes(plan, batch, sup_d, nosup_d, host_programs, after_sampling, before_prior, after_encode)
        ques_both = None
        if n_sup:
            program = batch["program"].to(dev)
            prog_sup = program[sup_d]
        if n_sup and n_nosup:
            # ONE encoder pass over the questions in the order [unsupervised rows, supervised rows]: the two decodes then
            # take the halves of its state as views (Seq2SeqBase.split_rows; rows are independent) and the reconstructor
            # its targets as they are -- gathering the two row sets from a pass in batch order cost 9 small launches forward
            # and 10 backward (six index_select, two index, a cat; four zero-fill + index_add_, two adds)
            ques_both = question.index_select(0, torch.cat((nosup_d, sup_d)))
            ques_nosup, ques_sup = ques_both[:n_nosup], ques_both[n_nosup:]
            state_nosup, state_sup = self.pg.split_rows(self.pg.encode(ques_both), n_nosup)
        elif n_sup:
            ques_sup = question[sup_d]
            state_sup = self.pg.encode(ques_sup)
        else:
            ques_nosup = question[nosup_d]
            state_nosup = self.pg.encode(ques_nosup)
        if after_encode is not None:
            out["after_encode"] = after_encode()
        paired = False
        group = None  # (the generator's prepared decodes and their launch: their graph node is created with the reconstructor's)
        if n_nosup:
            prep_s = None
            if n_sup and dev.type == "cuda":
                # the generator's sampling decode and its supervised (teacher-forced) decode start from the same encoder
                # pass and are independent: one launch each way for both -- the persistent decoder kernels are latency
                # bound, so the supervised pass rides along for free while the sampling pass, which the step's critical
                # chain waits for, takes as long as it did alone (Seq2SeqBase.decode_prepare / decode_pair)
                from probnmn.modules.seq2seq_base import decode_pair, decode_pair_launch

                prep_s = self.pg.decode_prepare(state_nosup, None, "sampling")
                prep_t = self.pg.decode_prepare(state_sup, prog_sup) if prep_s is not None else None
                if prep_t is not None and self.group_decoder_backward:
                    # ... and BACKWARD the reconstructor's decode joins them (nothing flows from the reconstruction into the
                    # generator: the samples are discrete): the pair is launched now, without a graph node; the node that
                    # owns all three passes is created where the reconstructor decodes (decode_group below)
                    pre = decode_pair_launch(prep_s, prep_t)
                    group = (prep_s, prep_t, pre)
                    z = self.pg._trim_predictions(pre.outs[1])
                    paired = True
                elif prep_t is not None:
                    out["pg"], o_sup = decode_pair(prep_s, prep_t)
                    out["pg_sup_rows"] = o_sup["loss"]
                    paired = True
            if not paired:
                drawn = prep_s["meta"]["seed"] if prep_s is not None else None  # (a prepared pass has drawn its seed)
                out["pg"] = self.pg.decode(state_nosup, None, "sampling", seed=drawn)
            if group is None:
                z = out["pg"]["predictions"]
            out["programs"] = z
            if host_programs:
                out["programs_host"] = self._host_copy(z)
            if after_sampling is not None:
                out["after_sampling"] = after_sampling()
        # per-row losses; "qr_rows" = the sampled rows' reconstruction losses followed by the supervised rows'
        if n_sup and not paired:
            out["pg_sup_rows"] = self.pg.decode(state_sup, prog_sup, "sampling", need_predictions=False)["loss"]
        if group is not None:
            from probnmn.modules.seq2seq_base import decode_group

            prep_s, prep_t, pre = group
            source = _cat_padded(z, prog_sup)
            prep_q = self.qr.decode_prepare(self.qr.encode(source), ques_both)
            if prep_q is not None:
                out["pg"], o_sup, o_qr = decode_group([prep_s, prep_t, prep_q], pre, [0, 1, None])
            else:  # (shapes outside the fused decoder: the reconstructor goes its own way)
                out["pg"], o_sup = decode_group([prep_s, prep_t], pre, [0, 1])
                o_qr = self.qr.decode(self.qr.encode(source), ques_both, "sampling", False)
            out["pg_sup_rows"], out["qr_rows"] = o_sup["loss"], o_qr["loss"]
        elif n_sup and n_nosup:
            out["qr_rows"] = self.qr(_cat_padded(z, prog_sup), ques_both, "sampling", False)["loss"]
        elif n_sup:
            out["qr_rows"] = self.qr(prog_sup, ques_sup, "sampling", False)["loss"]
        elif reconstruct:
            out["qr_rows"] = self.qr(z, ques_nosup, "sampling", False)["loss"]
        out["n_nosup"], out["n_sup"] = n_nosup, n_sup
        if before_prior is not None and n_nosup:
            out["before_prior"] = before_prior(out["programs_host"])
        if n_nosup and prior:
            with torch.no_grad():  # frozen model whose output only enters the detached reward
                out["prior"] = self.prior(z, need_predictions=False)["loss"]
        return out


    @staticmethod
    def _split_means(p) -> None:
        """The reference's separate terms from the batched passes' rows (the unfused path: CPU tensors, the joint
        "baseline" objective): p["qr"] = the sampled rows, p["pg_sup"] / p["qr_sup"] = the supervised rows' means."""
        n, m = p.get("n_nosup", 0), p.get("n_sup", 0)
        if "qr_rows" in p:
            if n:
                p["qr"] = p["qr_rows"][:n]
            if m:
                p["qr_sup"] = p["qr_rows"][n:].mean()
        if "pg_sup_rows" in p:
            p["pg_sup"] = p["pg_sup_rows"].mean()

    def _fused_objective(self, p, nmn_rows, w_sup, w_nosup, alpha, gamma):
        """(J, detached statistics) through ``pnmn_joint_objective`` -- one launch forward, one multiply backward."""
        n, m = p["n_nosup"], p["n_sup"]
        return self.elbo.objective(p["pg"]["loss"] if n else None, p["qr_rows"], p.get("prior"), nmn_rows,
                                   p.get("pg_sup_rows"), w_nosup, w_sup, alpha, gamma, n, m)


class QuestionCodingStep(_TrainerBase):
    def __init__(self, program_generator, question_reconstructor, program_prior, objective: str = "ours",
                 alpha: float = 100.0, beta: float = 0.1, delta: float = 0.99, lr: float = 1e-3,
                 weight_decay: float = 0.0, lr_gamma: float = 0.5, lr_patience: int = 3):
        if objective not in ("ours", "baseline"):
            raise ValueError("objective must be 'ours' or 'baseline'")
        self.pg, self.qr, self.prior = program_generator, question_reconstructor, program_prior
        self.objective, self.alpha = objective, alpha
        self.prior.eval()
        self.elbo = QuestionCodingElbo(self.pg, self.qr, self.prior, beta=beta, baseline_decay=delta)
        self.models = {"program_generator": self.pg, "question_reconstructor": self.qr}
        self.optimizer = self._make_optimizer([self.pg, self.qr], lr, weight_decay)
        self._init_schedule(lr_gamma, lr_patience)
        self.iteration = 0

    def step(self, batch: Dict[str, torch.Tensor]) -> Dict[str, Any]:
        self.optimizer.zero_grad()
        for m in (self.pg, self.qr):
            if not m.training:  # (Module.train() walks every submodule: 0.2-0.5 ms per model and step)
                m.train()
        dev = batch["question"].device
        sup, nosup = _split_supervision(batch["supervision"])
        sup_d, nosup_d = _index_to_device(sup, dev), _index_to_device(nosup, dev)
        ours = self.objective == "ours"
        p = self._seq2seq_passes(batch, sup_d, nosup_d, supervised=True, sampled=ours, prior=True)
        out: Dict[str, Any] = {}
        loss = torch.zeros((), device=dev)
        # Data parallel: both weights are computed (one collective each) and the REINFORCE baseline is
        # synchronised on EVERY rank, whatever this rank's shard holds -- a rank without supervised (or
        # without unsupervised) rows must issue the same sequence of collectives as the others.
        w_sup, w_nosup = _dp_weight(sup.numel(), dev), _dp_weight(nosup.numel(), dev)
        if ours and dev.type == "cuda" and "qr_rows" in p:
            loss, stats = self._fused_objective(p, None, w_sup, w_nosup, self.alpha, 0.0)
            if p["n_sup"]:
                out["loss"] = {k: stats[k] for k in ("program_generation_gt", "question_reconstruction_gt")}
            if p["n_nosup"]:
                out["elbo"] = {k: stats[k] for k in ("reconstruction_likelihood", "kl_divergence", "elbo", "reinforce_reward")}
                out["programs"] = p["programs"]
            self._finish(loss)
            out["objective"] = loss.detach()
            return out
        self._split_means(p)
        if "pg_sup" in p:
            loss = loss + w_sup * (self.alpha if ours else 1.0) * (p["pg_sup"] + p["qr_sup"])
            out["loss"] = {"program_generation_gt": p["pg_sup"].detach(), "question_reconstruction_gt": p["qr_sup"].detach()}
        if "pg" in p:
            elbo_out = self.elbo.combine(p["pg"]["loss"], p["qr"], p["prior"])
            loss = loss - w_nosup * elbo_out["elbo"]
            out["elbo"] = {k: v.detach() for k, v in elbo_out.items()}
            out["programs"] = p["programs"]
        elif ours and parallel.world() > 1:
            self.elbo._reinforce.idle(dev)
        self._finish(loss)
        out["objective"] = loss.detach()
        return out


class JointTrainingStep(_TrainerBase):
    def __init__(self, program_generator, question_reconstructor, program_prior, nmn, objective: str = "ours",
                 alpha: float = 100.0, beta: float = 0.1, gamma: float = 1.0, delta: float = 0.99,
                 lr: float = 1e-6, weight_decay: float = 0.0, lr_gamma: float = 0.5, lr_patience: int = 3):
        if objective not in ("ours", "baseline"):
            raise ValueError("objective must be 'ours' or 'baseline'")
        self.pg, self.qr, self.prior, self.nmn = program_generator, question_reconstructor, program_prior, nmn
        self.objective, self.alpha, self.gamma = objective, alpha, gamma
        self.prior.eval()
        self.elbo = JointTrainingElbo(self.pg, self.qr, self.prior, self.nmn, beta=beta, gamma=gamma,
                                      baseline_decay=delta, objective=objective)
        self.models = {"program_generator": self.pg, "question_reconstructor": self.qr, "nmn": self.nmn}
        self.optimizer = self._make_optimizer([self.pg, self.qr, self.nmn], lr, weight_decay)
        self._init_schedule(lr_gamma, lr_patience)
        self.iteration = 0
        self.blocked_seconds = 0.0  # host time spent waiting for the sampled programs (diagnostic, bench.py)
        # the NMN on its own stream beside the seq2seq passes (PNMN_NMN_STREAM=0: everything on one stream)
        self.nmn_stream = os.environ.get("PNMN_NMN_STREAM", "1") != "0"
        # (round 2 kept batches beyond 320 sampled rows on one stream: "either side fills the chip on its own".  Measured
        # again in round 3, profiles/ab/r04i_ab.txt / r04j_ab.txt, one box each: 512 questions 19.37 -> 17.8-18.9 ms,
        # 768: 25.55 -> 23.8, 1024: 32.3-32.4 -> 30.9-31.1 -- the deep program levels' launches of a few dozen items no
        # longer have the chip to themselves.  The attribute stays for A/B.)
        self.nmn_stream_max_rows = 1 << 30
        # the module programs are scheduled and launched between the reconstructor and the prior pass (the prior then runs
        # beside them) -- up to 511 questions; from 512 on after ALL seq2seq passes are issued: 128 questions 7.20-7.27 ms
        # against 7.28-7.42, 256: equal, 512: 17.92 against 17.80, 1024: 30.96 against 30.48-30.55 (the host's planning
        # of ~500 programs would otherwise hold the prior pass back; profiles/ab/r04k_ab.txt, r04l_ab.txt).
        # (None: by batch size; True / False fixes it.)
        self.trunk_before_prior = None
        # CUs the trunk's conv launches are cut for while it shares the chip with the seq2seq passes (side stream).  Up to
        # 128 questions: what their multi-CU kernels leave free -- eight workgroups per 16-row tile, one per CU: 224 at 64
        # questions (5.43-5.54 ms against 5.58-5.64 at 256), 192 at 128 (7.06-7.14 against 7.37-7.42; best of 160-256;
        # profiles/ab/r03x_ab.txt, r03z_ab.txt).  Beyond: the whole chip -- the multi-CU kernels then take (nearly) all of
        # it whenever they run, and the convs run between them: 256 questions 10.28-10.38 ms at 256 against 10.53-10.56 at
        # 192, 512: 17.8-18.9 / 18.5, 1024: 31.0-31.1 against 31.4-31.5 at 224 and 31.8 at 208 (profiles/ab/r04j_ab.txt).
        # (0: by batch size; a number fixes it.)
        self.shared_conv_cus = int(os.environ.get("PNMN_SHARED_CONV_CUS", "0"))  # (env: A/B aid)
        # (the same for the weight-gradient launches -- at most that many persistent workgroups -- measured at 128
        # questions: 192 -> 7.22 ms, 160 -> 7.8, unbounded 7.1-7.27: off by default, profiles/ab/r04a_ab.txt)
        self.shared_wgrad_cus = 0
        # When the stem (side stream) goes out.  0: first thing in the step (it then never waits for anything) -- below 256
        # questions.  2: issued behind the generator's encoder pass AND made to wait for it on the GPU -- from 256 questions
        # on: the encoder's multi-CU kernels otherwise become resident one workgroup at a time behind the stem's 1 ms conv
        # workgroups (encoder 0.9 -> 3.2 ms beside the stem at 1024 questions), and the sampled programs the module
        # programs wait for arrive that much later: 30.77-30.82 -> 30.43-30.49 ms, 512 questions 17.66 -> 17.31, 256: 10.30 -> 10.07
        # (profiles/ab/r04t_ab.txt, r04u_ab.txt; 128 questions: no difference).  1: issued behind the encoder pass without the wait (measured at 128 questions, r03f_ab.txt:
        # 7.85-7.89 against 7.83-7.92 ms, and at 1024: 30.78-30.96 -- no difference).  (None: by batch size.)
        self.stem_after_encode = None
        # the NMN's share of clamp + Adam on the NMN's stream, beside the next iteration's first passes (see _finish)
        self.overlap_optimizer = os.environ.get("PNMN_OVERLAP_OPTIMIZER", "0") != "0"
        self._side = None

    def _nmn_stream(self, dev) -> "torch.cuda.Stream":
        if self._side is None or self._side.device != dev:
            self._side = shared_stream(dev, "nmn trunk")
        return self._side

    def step(self, batch: Dict[str, torch.Tensor]) -> Dict[str, Any]:
        self.optimizer.zero_grad()
        for m in (self.pg, self.qr, self.nmn):
            if not m.training:  # (Module.train() walks every submodule: 0.2-0.5 ms per model and step)
                m.train()
        self.nmn.report_batch_metrics = False
        dev = batch["question"].device
        sup, nosup = _split_supervision(batch["supervision"])
        sup_d, nosup_d = _index_to_device(sup, dev), _index_to_device(nosup, dev)
        if nosup.numel() == 0 and parallel.world() == 1:
            raise ValueError("joint training needs at least one example without program supervision in the batch")
        ours = self.objective == "ours"
        # Data parallel: every rank issues the same collectives in the same order (both loss weights, the
        # REINFORCE baseline, the gradient all-reduces) whatever its shard holds; a shard without
        # unsupervised rows contributes zeros to the terms it has no rows for.
        w_sup, w_nosup = _dp_weight(sup.numel(), dev), _dp_weight(nosup.numel(), dev)
        out: Dict[str, Any] = {"loss": {}}
        if nosup.numel():
            answers = batch["answer"][nosup_d]
            main = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
            side = self._nmn_stream(dev) if (self.nmn_stream and main is not None
                                             and nosup.numel() <= self.nmn_stream_max_rows) else None
            # conv launches of a trunk that shares the chip are cut for the CUs it can count on (engine.conv_cus)
            engine = getattr(self.nmn, "engine", None)
            if engine is not None:
                rows = int(batch["question"].size(0))
                # (28x28 maps: four band units per item, launches four times as large -- the whole chip, as for the large
                # batches: 35.6-36.1 ms at 256 against 36.2-38.0 at 192, profiles/ab/r04r_c5.txt)
                free = self.shared_conv_cus or shared_conv_cus(rows, getattr(engine, "banded", False))
                engine.conv_cus = free if side is not None else 0
                engine.wgrad_cus = self.shared_wgrad_cus if side is not None else 0
            if side is not None:
                # The NMN runs on its own stream, beside the seq2seq passes: its stem needs no programs and
                # starts at once (next to the generator's encoder and sampling decode); its module programs
                # and classifier run next to the reconstructor / prior / supervised passes, and autograd
                # replays each side's backward on the stream its forward ran on, so the two backward passes
                # overlap as well.  Only this build's trunk kernels go to the side stream -- none of them waits
                # for another workgroup, so they always drain; the recurrent multi-CU kernels AND the library
                # GEMMs (fully connected layers included) stay on the main stream, one after the other: two
                # kernels that each wait for their own not-yet-resident workgroups can starve each other of
                # CUs forever (that is what stalled the side-stream experiment of round 1).
                side.wait_stream(main)  # the batch and the index tensors were produced on the main stream
                images, rows_d = _image_rows(batch["image"], nosup, nosup_d)
                token = {}

                stem_mode = (self.stem_after_encode if self.stem_after_encode is not None
                             else stem_waits_for_encoder(int(batch["question"].size(0))))

                def launch_stem():
                    if stem_mode == 2:  # (the stem also WAITS for the encoder pass on the GPU)
                        side.wait_stream(main)
                    # Beside the stem runs the generator's encoder, whose workgroups must ALL be resident before its first step:
                    # as a two-layer wavefront launch (probnmn.runtime.seq_plan) it needs twice the CUs a single layer does, and
                    # a stem cut for 192 CUs made it wait for the stem's 0.8 ms workgroups to drain (1.05 ms for the 0.2 ms
                    # launch, profiles/r06i_b128_timeline.txt).  The stem is needed ~1.5 ms later: it takes what is left.
                    budget = engine.conv_cus if engine is not None else 0
                    plan_wgs = self._planned_encoder_workgroups(batch, sup, nosup)
                    if engine is not None and budget and plan_wgs and not stem_mode:
                        engine.conv_cus = max(64, min(budget, 256 - plan_wgs))
                    with torch.cuda.stream(side):
                        # (the unsupervised examples' features -- 0.8 MB each -- are read through the row index by the
                        # layout kernel: no gathered copy)
                        token["started"] = self.nmn.begin(images, rows=rows_d)
                    if engine is not None:
                        engine.conv_cus = budget

                if not stem_mode:
                    launch_stem()

                def launch_trunk(programs_host):
                    # Between the reconstructor pass and the prior pass: by now the sampled programs are on the
                    # host, and the main stream has the reconstructor to work on while the host compiles and
                    # schedules them (~1 ms) and launches the trunk; the prior pass, issued afterwards, and the
                    # trunk then run side by side instead of one after the other.
                    host, copied = programs_host
                    t0 = time.perf_counter()
                    copied.synchronize()  # waits for the sampling decode only, not for the work queued after it
                    self.blocked_seconds += time.perf_counter() - t0
                    return self.nmn.forward_trunk(images, host, started=token["started"], trunk_stream=side, rows=rows_d)

                before_prior = (self.trunk_before_prior if self.trunk_before_prior is not None
                                else trunk_before_prior(int(batch["question"].size(0))))
                p = self._seq2seq_passes(batch, sup_d, nosup_d, supervised=ours, sampled=True, prior=ours,
                                         reconstruct=ours, host_programs=True,
                                         before_prior=launch_trunk if before_prior else None,
                                         after_encode=launch_stem if stem_mode else None)
                started = token["started"]
            else:
                images, rows_d = _image_rows(batch["image"], nosup, nosup_d)
                # one stream: the stem is queued right behind the sampling decode and keeps the GPU busy
                # (with the reconstructor / prior passes) while the host schedules the sampled programs
                p = self._seq2seq_passes(batch, sup_d, nosup_d, supervised=ours, sampled=True, prior=ours,
                                         reconstruct=ours, host_programs=True, after_sampling=lambda: self.nmn.begin(images, rows=rows_d))
                started = p["after_sampling"]
            if "before_prior" in p:
                nmn_out = self.nmn.forward_head(p["before_prior"], answers)
            else:
                programs_host, copied = p["programs_host"]
                t0 = time.perf_counter()
                copied.synchronize()  # waits for the sampling decode only, not for the work queued after it
                self.blocked_seconds += time.perf_counter() - t0
                nmn_out = self.nmn(images, programs_host, answers, started=started, trunk_stream=side, rows=rows_d)
            if ours and dev.type == "cuda":
                loss, stats = self._fused_objective(p, nmn_out["loss"], w_sup, w_nosup, self.alpha, self.gamma)
                _hip.mark("objective combined")
                out["loss"]["nmn"] = stats["nmn_loss"]
                out["elbo"] = {k: stats[k] for k in ("reconstruction_likelihood", "kl_divergence", "elbo", "reinforce_reward")}
                out["programs"] = p["programs"]
                if p["n_sup"]:
                    out["loss"]["program_generation_gt"] = stats["program_generation_gt"]
                    out["loss"]["question_reconstruction_gt"] = stats["question_reconstruction_gt"]
                self._finish(loss)
                out["objective"] = loss.detach()
                return out
            self._split_means(p)
            elbo_out = self.elbo.combine(p["pg"]["loss"], p.get("qr"), p.get("prior"), nmn_out)
            _hip.mark("elbo combined")
            nmn_loss = elbo_out.pop("nmn_loss")
            loss = w_nosup * (self.gamma * nmn_loss - elbo_out["elbo"])
            out["loss"]["nmn"] = nmn_loss.detach()
            out["elbo"] = {k: v.detach() for k, v in elbo_out.items()}
            out["programs"] = p["programs"]
        else:
            p = self._seq2seq_passes(batch, sup_d, nosup_d, supervised=ours, sampled=False, prior=False)
            self._split_means(p)
            self.elbo._reinforce.idle(dev)
            for a in self.optimizer.arenas:  # no NMN backward on this rank, which is what zeroes them
                a.grad.zero_()
            loss = torch.zeros((), device=dev)
        if "pg_sup" in p:
            loss = loss + w_sup * self.alpha * (p["pg_sup"] + p["qr_sup"])
            out["loss"]["program_generation_gt"] = p["pg_sup"].detach()
            out["loss"]["question_reconstruction_gt"] = p["qr_sup"].detach()
        self._finish(loss)
        out["objective"] = loss.detach()
        return out
