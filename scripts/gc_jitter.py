import gc, os, sys, time
ROOT = "/root/repo"
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
batch = bench.device_batch(vocab, 1024, 1000, dev)
bench.fit_program_generator(pg, vocab, batch, dev, 1500, 0.95)
step = JointTrainingStep(pg, qr, prior, nmn, **bench.JOINT)
for _ in range(3): step.step(batch)
def run(n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ts = []
    for _ in range(n):
        t1 = time.perf_counter(); step.step(batch); ts.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, max(ts) * 1e3, min(ts) * 1e3
for mode in ("gc on", "gc off", "gc on", "gc off"):
    if mode == "gc off": gc.collect(); gc.disable()
    else: gc.enable()
    r = [run() for _ in range(3)]
    print(mode, ["%.1f (host step %.1f..%.1f)" % x for x in r], flush=True)
print("alloc retries", torch.cuda.memory_stats()["num_alloc_retries"], "cudaMalloc calls", torch.cuda.memory_stats()["num_device_alloc"])
