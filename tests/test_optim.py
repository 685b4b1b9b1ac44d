"""ClampAdam as the reference's optimizer object (reference: probnmn/trainers/_trainer.py:103-130):
a torch.optim.Optimizer that ReduceLROnPlateau can drive and whose state_dict is torch.optim.Adam's.
The CPU tests cover the host logic (groups, scheduler, state layout); the fused update itself is a HIP
kernel and is tested on the MI355X."""
import copy

import pytest
import torch
from torch import nn

from probnmn.optim import ClampAdam


def _model():
    torch.manual_seed(0)
    return nn.Sequential(nn.Linear(6, 5), nn.Tanh(), nn.Linear(5, 3))


def test_is_an_optimizer_with_one_group_and_adam_state_layout():
    m = _model()
    opt = ClampAdam(m.parameters(), lr=3e-4, weight_decay=0.01)
    assert isinstance(opt, torch.optim.Optimizer)
    assert len(opt.param_groups) == 1 and opt.param_groups[0]["lr"] == 3e-4
    assert opt.param_groups[0]["betas"] == (0.9, 0.999) and opt.param_groups[0]["eps"] == 1e-8
    sd = opt.state_dict()
    ref = torch.optim.Adam(_model().parameters(), lr=3e-4, weight_decay=0.01)
    assert sd["param_groups"][0]["params"] == ref.state_dict()["param_groups"][0]["params"]
    assert sd["state"] == ref.state_dict()["state"] == {}  # (no gradient so far: no state, as torch.optim.Adam)
    for p in m.parameters():  # the buffers a step will update exist and alias the state
        st = opt.state[p]
        assert set(st) == {"step", "exp_avg", "exp_avg_sq"} and st["exp_avg"].shape == p.shape and float(st["step"]) == 0.0
    with pytest.raises(ValueError):
        opt.add_param_group({"params": [nn.Parameter(torch.zeros(2))]})


def test_reduce_lr_on_plateau_drives_it():
    """ReduceLROnPlateau(mode="max", factor=LR_GAMMA, patience=LR_PATIENCE, threshold=1e-3), as _trainer.py:110-118."""
    opt = ClampAdam(_model().parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="max", factor=0.5, patience=2, threshold=1e-3)
    for metric in (0.50, 0.5001, 0.5002, 0.5003):  # flat within the threshold: patience 2 -> halves on the 4th
        sched.step(metric)
    assert opt.param_groups[0]["lr"] == pytest.approx(5e-4) and opt.lr == pytest.approx(5e-4)
    sched.step(0.9)
    assert opt.lr == pytest.approx(5e-4)
    # the scheduler's own state round-trips next to the optimizer's, as the reference's checkpoints hold both
    sched2 = torch.optim.lr_scheduler.ReduceLROnPlateau(ClampAdam(_model().parameters(), lr=1e-3), mode="max", factor=0.5, patience=2)
    sched2.load_state_dict(sched.state_dict())
    assert sched2.best == sched.best


def test_state_dict_round_trips_with_torch_adam():
    """Optimizer state written by torch.optim.Adam (what a reference checkpoint holds) loads into
    ClampAdam and back, moments and step included."""
    m = _model()
    adam = torch.optim.Adam(m.parameters(), lr=2e-3)
    for _ in range(3):
        adam.zero_grad()
        m(torch.randn(4, 6)).pow(2).sum().backward()
        adam.step()
    sd = copy.deepcopy(adam.state_dict())
    m2 = copy.deepcopy(m)
    opt = ClampAdam(m2.parameters(), lr=1.0)
    opt.load_state_dict(sd)
    assert opt.step_count == 3 and opt.lr == 2e-3
    for (p, (mom, var)), st in zip(zip(opt.loose, opt._loose_state), sd["state"].values()):
        assert torch.equal(mom, st["exp_avg"]) and torch.equal(var, st["exp_avg_sq"])
        assert opt.state[p]["exp_avg"].data_ptr() == mom.data_ptr()  # state aliases the buffers the kernel updates
    back = opt.state_dict()
    adam2 = torch.optim.Adam(copy.deepcopy(m).parameters(), lr=1.0)
    adam2.load_state_dict(back)
    for a, b in zip(adam2.state_dict()["state"].values(), sd["state"].values()):
        assert float(a["step"]) == float(b["step"]) == 3.0
        assert torch.equal(a["exp_avg"], b["exp_avg"]) and torch.equal(a["exp_avg_sq"], b["exp_avg_sq"])


@pytest.mark.gpu
def test_fused_step_follows_the_scheduler_and_resumes_from_a_checkpoint():
    """On the MI355X: (1) ClampAdam == clamp + torch.optim.Adam over an NMN arena + loose tensors;
    (2) after ReduceLROnPlateau halves the lr the next fused step uses it; (3) a state_dict saved mid-run
    restores into a fresh optimizer (and into torch.optim.Adam) and both continue identically."""
    from probnmn.models.nmn import NeuralModuleNetwork
    from probnmn.vocabulary import Vocabulary

    dev = torch.device("cuda:0")
    vocab = Vocabulary.clevr()
    torch.manual_seed(1)
    net = NeuralModuleNetwork(vocab, class_projection_channels=128, classifier_linear_size=32).to(dev)
    arena = net.engine.ensure_arena()
    names = [n for n, _ in net.named_parameters()]
    opt = ClampAdam(net.parameters(), arenas=[arena], lr=1e-2, weight_decay=0.01)
    ref_params = [p.detach().cpu().clone().requires_grad_(True) for p in net.parameters()]
    ref = torch.optim.Adam(ref_params, lr=1e-2, weight_decay=0.01)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="max", factor=0.5, patience=0, threshold=1e-3)
    sched_ref = torch.optim.lr_scheduler.ReduceLROnPlateau(ref, mode="max", factor=0.5, patience=0, threshold=1e-3)
    g = torch.Generator().manual_seed(2)

    arena_index = {n: i for i, n in enumerate(arena.names)}

    def one_step(optimizers, unused=()):
        """``unused``: parameter names this step's "batch" does not reach -- no gradient from backward.  The reference side
        is torch 1.4.0's bookkeeping (zero_grad zeroes in place): None before a parameter's first gradient, zeros after."""
        grads = [torch.randn(p.shape, generator=g) * 4 for p in ref_params]  # |g| > 5 occurs: the clamp matters
        arena.grad.zero_()
        for p, gr in zip(net.parameters(), grads):
            p.grad = None
        arena.attach_grads()
        arena.touched[:] = False
        for n, p, gr in zip(names, net.parameters(), grads):
            if n in unused:
                continue
            if n in arena_index:
                arena.touched[arena_index[n]] = True  # (what NMNEngine.run_backward records)
                p.grad.copy_(gr.to(dev))
            else:
                p.grad = gr.to(dev)
        for n, p, gr in zip(names, ref_params, grads):
            if n in unused:
                p.grad = None if p.grad is None else torch.zeros_like(p)
            else:
                p.grad = gr.clamp(-5, 5)
        for o in optimizers:
            o.step()

    def check(tag, tol=2e-6):
        for n, p, r in zip(names, net.parameters(), ref_params):
            torch.testing.assert_close(p.detach().cpu(), r.detach(), rtol=1e-5, atol=tol, msg=lambda m: "%s %s: %s" % (tag, n, m))

    # per-parameter step counts: two modules and a fully connected layer join at the second step (their first Adam step is a
    # full lr there), one module sits the third step out (its momentum goes on)
    late = {n for n in names if n.startswith(("filter_color[red].", "relate[left].", "classifier.6."))}
    idle = {n for n in names if n.startswith("filter_size[small].")}
    assert len(late) >= 10 and len(idle) >= 4
    one_step([opt, ref], unused=late)
    one_step([opt, ref])
    check("two steps, some parameters from the second on")
    assert {float(ref.state[p]["step"]) for n, p in zip(names, ref_params) if n in late} == {1.0}
    sd = opt.state_dict()["state"]
    assert [float(sd[i]["step"]) for i in range(len(names))] == [float(ref.state[p]["step"]) for p in ref_params]
    one_step([opt, ref], unused=idle)
    check("a module without gradient keeps its momentum")
    sched.step(0.5), sched_ref.step(0.5)
    sched.step(0.5), sched_ref.step(0.5)  # no improvement, patience 0 -> lr halves
    assert opt.lr == pytest.approx(5e-3) == ref.param_groups[0]["lr"]
    one_step([opt, ref])
    check("after the lr change")

    # checkpoint mid-run: ClampAdam -> fresh ClampAdam, and -> torch.optim.Adam
    sd = copy.deepcopy(opt.state_dict())
    assert [float(sd["state"][i]["step"]) for i in range(len(names))] == [float(ref.state[p]["step"]) for p in ref_params]
    assert sorted({float(st["step"]) for st in sd["state"].values()}) == [3.0, 4.0]
    opt2 = ClampAdam(net.parameters(), arenas=[arena], lr=123.0)
    opt2.load_state_dict(sd)
    ref2 = torch.optim.Adam(ref_params, lr=123.0)
    ref2.load_state_dict({"state": {k: {kk: vv.cpu() for kk, vv in v.items()} for k, v in sd["state"].items()},
                          "param_groups": sd["param_groups"]})
    assert opt2.lr == pytest.approx(5e-3) and opt2.step_count == 4
    one_step([opt2, ref2])
    check("resumed")

    # a model moved after the optimizer was built is detected, not silently ignored
    net.stem[0].weight.data = net.stem[0].weight.data.clone()
    with pytest.raises(Exception, match="arena"):
        opt2.step()


def test_loading_a_torch_adam_state_keeps_the_group_options():
    """torch's load_state_dict REPLACES param_groups with the saved ones; a reference checkpoint's
    Adam group has no 'clamp' key (step() would raise KeyError) and may carry options the fused update does not
    implement -- those are refused, not silently ignored."""
    m = _model()
    adam = torch.optim.Adam(m.parameters(), lr=2e-3)
    m(torch.randn(4, 6)).pow(2).sum().backward()
    adam.step()
    opt = ClampAdam(copy.deepcopy(m).parameters(), lr=1.0, clamp=5.0)
    opt.load_state_dict(copy.deepcopy(adam.state_dict()))
    g = opt.param_groups[0]
    assert g["clamp"] == 5.0 and g["lr"] == 2e-3 and g["betas"] == (0.9, 0.999) and opt._check_full
    for flag in ("amsgrad", "maximize"):
        sd = copy.deepcopy(adam.state_dict())
        sd["param_groups"][0][flag] = True
        with pytest.raises(ValueError, match=flag):
            ClampAdam(copy.deepcopy(m).parameters(), lr=1.0).load_state_dict(sd)


@pytest.mark.gpu
def test_steps_after_loading_a_reference_adam_state():
    """The resume path _base.load_state_dict advertises: optimizer state written by torch.optim.Adam (a reference
    checkpoint) -> ClampAdam -> ONE fused step == clamp + torch Adam's next step."""
    dev = torch.device("cuda:0")
    m = _model()
    adam = torch.optim.Adam(m.parameters(), lr=2e-3)
    for _ in range(2):
        adam.zero_grad()
        m(torch.randn(4, 6)).pow(2).sum().backward()
        adam.step()
    m_dev = copy.deepcopy(m).to(dev)
    opt = ClampAdam(m_dev.parameters(), lr=1.0, clamp=5.0)
    opt.load_state_dict(copy.deepcopy(adam.state_dict()))
    grads = [torch.randn(p.shape) * 4 for p in m.parameters()]
    for p, q, gr in zip(m.parameters(), m_dev.parameters(), grads):
        p.grad, q.grad = gr.clamp(-5, 5), gr.to(dev)
    adam.step()
    opt.step()
    assert opt.step_count == 3
    for p, q in zip(m.parameters(), m_dev.parameters()):
        torch.testing.assert_close(q.detach().cpu(), p.detach(), rtol=1e-5, atol=2e-6)
