"""Host time of a joint-training step by phase (wall clock of the Python calls that enqueue the work; the GPU
runs behind).  usage: python scripts/host_phases.py [batch]"""
import os, sys, time
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
batch = bench.device_batch(vocab, B, 1000, dev)
bench.fit_program_generator(pg, vocab, batch, dev, 1500, 0.95)
step = JointTrainingStep(pg, qr, prior, nmn, **bench.JOINT)
acc = {}
def wrap(obj, name, label):
    fn = getattr(obj, name)
    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
    setattr(obj, name, timed)
wrap(pg, "encode", "pg.encode")
wrap(pg, "decode", "pg.decode (sampling + supervised)")
wrap(qr, "forward", "qr pass"); qr_call = qr.__call__
wrap(prior, "forward", "prior pass")
wrap(nmn, "begin", "nmn.begin (stem)")
wrap(nmn, "forward", "nmn.forward (compile, plan, launches, FC, loss)")
wrap(step.elbo, "combine", "elbo.combine")
wrap(step.optimizer, "step", "optimizer.step")
import probnmn.parallel as par
orig_backward = torch.Tensor.backward
def timed_backward(self, *a, **k):
    t0 = time.perf_counter(); orig_backward(self, *a, **k); acc["loss.backward (autograd thread issues all backward launches)"] = acc.get("loss.backward (autograd thread issues all backward launches)", 0.0) + time.perf_counter() - t0
torch.Tensor.backward = timed_backward
for _ in range(5): step.step(batch)
torch.cuda.synchronize(); acc.clear()
N = 20
t0 = time.perf_counter()
for _ in range(N): step.step(batch)
host = time.perf_counter() - t0
torch.cuda.synchronize()
total = time.perf_counter() - t0
print("B=%d: %.2f ms/step wall, host loop %.2f ms/step, blocked on sampled programs %.2f ms/step" % (B, total / N * 1e3, host / N * 1e3, step.blocked_seconds / (N + 5) * 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-70s %.2f ms" % (k, v / N * 1e3))
print("  %-70s %.2f ms" % ("(rest: split, gathers, weights, host copy, python)", (host - sum(acc.values())) / N * 1e3))
