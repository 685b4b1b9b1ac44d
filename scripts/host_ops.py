"""CPU-side op table of a joint-training step (torch profiler, CPU activity only): which ops cost host time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd")]
import torch
import bench
from probnmn.models import NeuralModuleNetwork, ProgramGenerator, ProgramPrior, QuestionReconstructor
from probnmn.trainers.joint_training import JointTrainingStep
from probnmn.vocabulary import Vocabulary
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
vocab = Vocabulary.clevr()
torch.manual_seed(0)
nmn = NeuralModuleNetwork(vocab).to(dev)
pg, qr, prior = ProgramGenerator(vocab).to(dev), QuestionReconstructor(vocab).to(dev), ProgramPrior(vocab, hidden_size=256).to(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
batch = bench.device_batch(vocab, B, 1000, dev)
bench.fit_program_generator(pg, vocab, batch, dev, 1500, 0.95)
step = JointTrainingStep(pg, qr, prior, nmn, **bench.JOINT)
for _ in range(5): step.step(batch)
torch.cuda.synchronize()
N = 10
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(N): step.step(batch)
    torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.self_cpu_time_total)
tot = sum(e.self_cpu_time_total for e in rows)
print("total self CPU %.2f ms/step over %d op kinds, %d calls/step" % (tot / N / 1e3, len(rows), sum(e.count for e in rows) // N))
for e in rows[:45]:
    print("%-60s calls/step %6.1f  self %.3f ms/step  total %.3f ms/step" % (e.key[:60], e.count / N, e.self_cpu_time_total / N / 1e3, e.cpu_time_total / N / 1e3))
