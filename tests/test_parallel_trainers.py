"""The trainers' data-parallel control flow with world_size 2 over gloo on CPU: shards whose supervised or
unsupervised subset is EMPTY on one rank must issue the same collectives as the other rank (no hang, no
mis-paired all-reduce) and the averaged gradients must equal the single-process gradients on the
concatenated batch (joint_training.py collectives were guarded on local subset sizes).

The model passes are replaced by small differentiable stand-ins (the real ones are HIP kernels and are
tested on the MI355X); everything else -- subset split, loss weights n_local * world / n_global,
REINFORCE baseline synchronisation, ELBO combination, gradient all-reduce order -- is the product code."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _Toy(torch.nn.Module):
    def __init__(self, seed, n=4):
        super().__init__()
        self.w = torch.nn.Parameter(torch.randn(n, generator=torch.Generator().manual_seed(seed)) * 0.5)
        self.big = torch.nn.Parameter(torch.randn(n, generator=torch.Generator().manual_seed(seed + 50)) * 0.5)
        self.report_batch_metrics = False

    def row_loss(self, x, offset):
        return (x @ self.w).pow(2) + (x @ self.big).pow(2) + offset

    def forward(self, images, programs, answers, started=None, trunk_stream=None, rows=None):  # the NMN stand-in
        return {"loss": self.row_loss(images if rows is None else images[rows], 0.25)}


class _Done:
    def synchronize(self):
        pass


def _make(kind, objective):
    sys.path.insert(0, os.path.join(ROOT, "probnmn-clevr_amd"))
    from probnmn import parallel
    from probnmn.trainers.joint_training import JointTrainingStep, QuestionCodingStep

    base = JointTrainingStep if kind == "joint" else QuestionCodingStep

    class ToyStep(base):
        def _seq2seq_passes(self, batch, sup_d, nosup_d, supervised, sampled, prior, reconstruct=True,
                            host_programs=False, after_sampling=None):
            x = batch["x"]
            out = {}
            n_sup = sup_d.numel() if supervised else 0
            n_nosup = nosup_d.numel() if sampled else 0
            if n_nosup:
                xs = x[nosup_d]
                out["pg"] = {"loss": self.pg.row_loss(xs, 0.5)}
                out["programs"] = torch.zeros(n_nosup, 3, dtype=torch.long)
                out["programs_host"] = (out["programs"], _Done())
                out["after_sampling"] = None
                if reconstruct:
                    out["qr"] = self.qr.row_loss(xs, 1.0)
                if prior:
                    out["prior"] = self.prior.row_loss(xs, 0.1).detach()
            if n_sup:
                xs = x[sup_d]
                out["pg_sup"] = self.pg.row_loss(xs, 0.0).mean()
                out["qr_sup"] = self.qr.row_loss(xs, 0.0).mean()
            return out

        def _make_optimizer(self, models, lr, weight_decay):
            opt = super()._make_optimizer(models, lr, weight_decay)
            opt.step = lambda: None  # (the fused update is a HIP kernel; gradients are what is checked)
            # the parameters named `big` take the early path: their all-reduce starts inside backward
            self._early = parallel.EarlyReducer([m.big for m in models])
            return opt

    pg, qr, prior, nmn = _Toy(1), _Toy(2), _Toy(3), _Toy(4)
    if kind == "joint":
        return ToyStep(pg, qr, prior, nmn, objective=objective, alpha=100.0, beta=0.1, gamma=1.0, delta=0.5), (pg, qr, nmn)
    return ToyStep(pg, qr, prior, objective=objective, alpha=100.0, beta=0.1, delta=0.5), (pg, qr)


def _batch(rows, supervision):
    g = torch.Generator().manual_seed(77)
    x = torch.randn(8, 4, generator=g)
    return {"x": x[rows], "question": x[rows], "image": x[rows], "answer": torch.zeros(len(rows), dtype=torch.long),
            "supervision": torch.tensor(supervision)[rows]}


def _run(kind, objective, supervision, rows):
    trainer, models = _make(kind, objective)
    grads = []
    for it in range(2):  # two iterations: the REINFORCE baseline of the first enters the second
        trainer.step(_batch(rows, supervision))
        grads.append([torch.zeros_like(p) if p.grad is None else p.grad.clone() for m in models for p in m.parameters()])
    return grads, float(trainer.elbo._reinforce._reinforce_baseline)


def _worker(rank, world, port, kind, objective, supervision, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = list(range(0, 3)) if rank == 0 else list(range(3, 8))
    grads, baseline = _run(kind, objective, supervision, rows)
    torch.save({"grads": grads, "baseline": baseline}, out_path + str(rank))
    dist.barrier()
    dist.destroy_process_group()


CASES = [
    # rank 0 = rows 0-2, rank 1 = rows 3-7
    ("joint", "ours", [1, 1, 1, 1, 0, 0, 1, 0]),          # rank 0 holds no unsupervised row
    ("joint", "ours", [1, 0, 1, 0, 0, 0, 0, 0]),          # rank 1 holds no supervised row
    ("joint", "baseline", [1, 1, 1, 0, 0, 1, 0, 0]),      # baseline objective: no supervised terms at all
    ("question_coding", "ours", [1, 1, 1, 1, 0, 0, 1, 0]),
    ("question_coding", "ours", [0, 0, 0, 1, 1, 0, 1, 0]),  # rank 0 holds no supervised row
]


@pytest.mark.parametrize("kind,objective,supervision", CASES)
def test_empty_subset_on_one_rank_matches_single_process(tmp_path, kind, objective, supervision):
    out = str(tmp_path / "r")
    mp.spawn(_worker, args=(2, _free_port(), kind, objective, supervision, out), nprocs=2, join=True)
    got0, got1 = torch.load(out + "0"), torch.load(out + "1")
    want, want_baseline = _run(kind, objective, supervision, list(range(8)))  # one process, the whole batch
    for it in range(2):
        for a, b, w in zip(got0["grads"][it], got1["grads"][it], want[it]):
            torch.testing.assert_close(a, b, rtol=0, atol=0)  # both ranks hold the same averaged gradient
            torch.testing.assert_close(a, w, rtol=1e-5, atol=1e-6)
    assert got0["baseline"] == pytest.approx(want_baseline, rel=1e-6, abs=1e-7)
    assert got1["baseline"] == pytest.approx(want_baseline, rel=1e-6, abs=1e-7)


def test_second_trainer_takes_over_the_early_hook():
    """Two trainers over the same parameter (module_training then joint_training on one NMN): only the
    newer reducer's hook stays live, so a gradient is never all-reduced twice (parallel.py:33)."""
    sys.path.insert(0, os.path.join(ROOT, "probnmn-clevr_amd"))
    from probnmn import parallel

    p = torch.nn.Parameter(torch.zeros(3))
    first = parallel.EarlyReducer([p])
    second = parallel.EarlyReducer([p])
    assert first.params == [] and second.params == [p]
    assert len(p._post_accumulate_grad_hooks) == 1
    second.remove()
    assert not p._post_accumulate_grad_hooks


def test_unarmed_reducer_starts_no_collective(monkeypatch):
    """the hook's owner starts collectives only while ITS trainer runs backward.  The older trainer
    stepping again (its optimizer still lists the parameter) must not leave an orphan all-reduce in the newer
    reducer, nor have the gradient summed twice."""
    sys.path.insert(0, os.path.join(ROOT, "probnmn-clevr_amd"))
    from probnmn import parallel

    calls = []

    class _Done:
        def wait(self):
            pass

    monkeypatch.setattr(parallel, "world", lambda: 2)
    monkeypatch.setattr(parallel.dist, "all_reduce", lambda t, op=None, async_op=False: (calls.append(t), _Done())[1])
    p = torch.nn.Parameter(torch.ones(3))
    old, new = parallel.EarlyReducer([p]), parallel.EarlyReducer([p])  # `new` took the hook over
    (p * 2.0).sum().backward()  # the OLD trainer's backward: nobody armed
    assert calls == [] and not new._pending
    parallel.all_reduce_gradients([], [p], early=old, average=False)  # old has no early parameters left
    assert len(calls) == 1  # reduced exactly once, as an ordinary loose tensor
    calls.clear()
    p.grad = None
    new.arm()
    (p * 2.0).sum().backward()
    assert len(calls) == 1 and id(p) in new._pending  # started from the hook
    parallel.all_reduce_gradients([], [p], early=new, average=False)
    assert len(calls) == 1 and not new._pending and not new.armed


def test_joint_step_schedule_rules_follow_the_batch():
    """The three batch-dependent rules of the joint step's two-stream schedule (measured thresholds)."""
    from probnmn.trainers import joint_training as jt

    assert [jt.shared_conv_cus(n, False) for n in (16, 64, 128, 129, 256, 1024)] == [248, 224, 192, 0, 0, 0]
    assert jt.shared_conv_cus(128, True) == 0  # 28x28 maps: four band units per item
    assert [jt.stem_waits_for_encoder(n) for n in (128, 255, 256, 1024)] == [0, 0, 2, 2]
    assert [jt.trunk_before_prior(n) for n in (128, 511, 512, 1024)] == [True, True, False, False]
