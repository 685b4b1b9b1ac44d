"""Which torch ops a joint-training step issues, by call site (forward: python stack; backward: autograd node)
and by input shapes.  usage: python scripts/op_sites.py [batch]"""
import os, sys, collections
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
N = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(N): step.step(batch)
    torch.cuda.synchronize()
ev = prof.events()
PKG = os.path.join("probnmn-clevr_amd", "probnmn")
sites = collections.Counter(); cpu = collections.Counter(); kern = collections.Counter(); dk = collections.Counter()
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU or e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::"):
        continue
    if not (e.name.startswith("aten::") or "Backward" in e.name or e.name.startswith("_") ):
        continue
    site = "?"
    for fr in (e.stack or []):
        if PKG in fr or "bench.py" in fr:
            site = fr.split(PKG + "/")[-1][:70]
            break
    if site == "?" and e.cpu_parent is not None:
        p = e.cpu_parent
        while p is not None and not ("Backward" in p.name or "autograd" in p.name): p = p.cpu_parent
        site = "bwd:" + (p.name[:60] if p is not None else "?")
    key = (e.name, site, str(e.input_shapes)[:60])
    sites[key] += 1
    cpu[key] += e.cpu_time_total
    nk = len(e.kernels)
    def allk(x):
        n = len(x.kernels); t = sum(k.duration for k in x.kernels)
        for c in x.cpu_children:
            a, b = allk(c); n += a; t += b
        return n, t
    n, t = allk(e)
    kern[key] += n; dk[key] += t
print("%-34s %-72s %-40s %7s %9s %8s %9s" % ("op", "site", "shapes", "n/step", "cpu us/st", "kern/st", "gpu us/st"))
for key, n in sorted(sites.items(), key=lambda kv: -cpu[kv[0]]):
    if kern[key] == 0 and cpu[key] / N < 20: continue
    print("%-34s %-72s %-40s %7.1f %9.1f %8.1f %9.1f" % (key[0][:34], key[1][:72], key[2][:40], n / N, cpu[key] / N, kern[key] / N, dk[key] / N))
