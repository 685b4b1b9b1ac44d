"""Host time of the pieces of NeuralModuleNetwork.forward/backward in a joint step (no device syncs added)."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd")]
import torch
import bench
from probnmn.models import NeuralModuleNetwork, ProgramGenerator, ProgramPrior, QuestionReconstructor
from probnmn.trainers.joint_training import JointTrainingStep
from probnmn.vocabulary import Vocabulary
from probnmn.runtime import engine as E, schedule as S
from probnmn import _hip

dev = torch.device("cuda:0")
vocab = Vocabulary.clevr()
torch.manual_seed(0)
nmn = NeuralModuleNetwork(vocab).to(dev)
pg, qr, prior = ProgramGenerator(vocab).to(dev), QuestionReconstructor(vocab).to(dev), ProgramPrior(vocab, hidden_size=256).to(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
batch = bench.device_batch(vocab, B, 1000, dev)
bench.fit_program_generator(pg, vocab, batch, dev, 1500, 0.95)
step = JointTrainingStep(pg, qr, prior, nmn, **bench.JOINT)
acc = collections.Counter()

def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label] += time.perf_counter() - t
    setattr(obj, name, g)

eng = nmn.engine
wrap(eng.scheduler, "plan", "scheduler.plan")
wrap(eng.scheduler, "arena_floats")
for n in ("compile", "begin_forward", "run_forward", "run_backward", "_run_forward_launches", "_flush_list", "_fixed_records"):
    if hasattr(eng, n): wrap(eng, n, "engine." + n)
wrap(E._Pack, "upload", "pack.upload")
wrap(eng.compiler, "compile_batch", "compiler.compile_batch")
wrap(eng.scheduler, "template_id", "scheduler.template_id")
wrap(_hip, "to_device", "_hip.to_device")
wrap(_hip, "small_to_device", "_hip.small_to_device")
wrap(nmn, "forward", "nmn.forward")
wrap(nmn, "begin", "nmn.begin")
for _ in range(6): step.step(batch)
torch.cuda.synchronize()
acc.clear()
N = 20
t0 = time.perf_counter()
for _ in range(N): step.step(batch)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N
print("B=%d  %.2f ms/step" % (B, wall * 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-32s %7.3f ms/step" % (k, v / N * 1e3))
