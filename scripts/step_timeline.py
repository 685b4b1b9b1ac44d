"""Where a joint-training step's time goes, WITHOUT a profiler attached: host time and GPU time (events on the
stream that is current at that point) at the entry and exit of the step's phases, averaged over steps.
usage: python scripts/step_timeline.py [batch] [steps]"""
import os, sys, time, collections, functools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd")]
import torch
import bench
from probnmn.models import NeuralModuleNetwork, ProgramGenerator, ProgramPrior, QuestionReconstructor
from probnmn.trainers.joint_training import JointTrainingStep
from probnmn.vocabulary import Vocabulary

dev = torch.device("cuda:0")
vocab = Vocabulary.clevr()
torch.manual_seed(0)
nmn = NeuralModuleNetwork(vocab).to(dev)
pg, qr, prior = ProgramGenerator(vocab).to(dev), QuestionReconstructor(vocab).to(dev), ProgramPrior(vocab, hidden_size=256).to(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 30
batch = bench.device_batch(vocab, B, 1000, dev)
bench.fit_program_generator(pg, vocab, batch, dev, 1500, 0.95)
step = JointTrainingStep(pg, qr, prior, nmn, **bench.JOINT)

marks = []  # (name, host time, event, stream id) of the current step


def mark(name):
    ev = torch.cuda.Event(enable_timing=True)
    s = torch.cuda.current_stream(dev)
    ev.record(s)
    marks.append((name, time.perf_counter(), ev, s.cuda_stream))


def wrap(obj, attr, name):
    f = getattr(obj, attr)

    @functools.wraps(f)
    def g(*a, **k):
        mark(name + " >")
        try:
            return f(*a, **k)
        finally:
            mark(name + " <")
    setattr(obj, attr, g)


wrap(pg, "encode", "pg.encode")
wrap(pg, "decode", "pg.decode")
wrap(qr, "forward", "qr")
wrap(prior, "forward", "prior")
wrap(nmn, "begin", "nmn.begin(stem)")
wrap(nmn, "forward", "nmn.forward (plan + trunk + head)")
wrap(nmn.engine, "run_forward_tokens", "trunk fwd")
wrap(nmn, "forward_trunk", "nmn.forward_trunk")
wrap(nmn, "forward_head", "nmn.forward_head")
wrap(step.elbo, "objective", "objective")
wrap(step, "_finish", "_finish")
wrap(nmn.engine, "run_backward", "trunk bwd")
wrap(step.optimizer, "step", "optimizer")
orig_backward = torch.Tensor.backward


def backward(self, *a, **k):
    mark("loss.backward >")
    r = orig_backward(self, *a, **k)
    mark("loss.backward <")
    return r


torch.Tensor.backward = backward
for _ in range(10):
    step.step(batch); marks.clear()
torch.cuda.synchronize()
FREE = "--free" in sys.argv  # no synchronisation between the steps: the steady state of a training loop
acc = collections.OrderedDict()
per_step = []
torch.cuda.synchronize()
t_all = time.perf_counter()
for _ in range(N):
    marks.clear()
    if not FREE:
        torch.cuda.synchronize()
    mark("step >")
    step.step(batch)
    mark("step <")
    if not FREE:
        torch.cuda.synchronize()
    per_step.append(list(marks))
torch.cuda.synchronize()
total = time.perf_counter() - t_all
for ms in per_step[N // 3:]:  # (free-running: the first steps still fill the pipeline)
    base_h, base_e = ms[0][1], ms[0][2]
    seen = collections.Counter()
    for name, h, ev, s in ms:
        seen[name] += 1
        key = "%s #%d" % (name, seen[name]) if seen[name] > 1 or name.startswith(("pg.decode", "trunk")) else name
        a = acc.setdefault(key, [0.0, 0.0, 0, s])
        a[0] += (h - base_h) * 1e3
        a[1] += base_e.elapsed_time(ev)
        a[2] += 1
main = per_step[0][0][3]
if FREE:
    print("batch %d, free running: %.2f ms per step; times relative to the host / the GPU reaching the step's start "
          "(GPU ms < host ms + lag means the GPU is behind the host there)" % (B, total / N * 1e3))
    lag = [ms[0][2] for ms in per_step]
    print("GPU start of a step after the previous step's GPU start: %.2f ms" % (sum(lag[i].elapsed_time(lag[i + 1]) for i in range(N // 3, N - 1)) / (N - 1 - N // 3)))
else:
    print("batch %d: %.2f ms per step with one synchronisation per step (a free-running loop overlaps the steps' ends and starts)" % (B, total / N * 1e3))
print("%-28s %10s %10s  %s" % ("point", "host ms", "GPU ms", "stream"))
for k, (h, g, n, s) in acc.items():
    print("%-28s %10.2f %10.2f  %s" % (k, h / n, g / n, "main" if s == main else "side"))
