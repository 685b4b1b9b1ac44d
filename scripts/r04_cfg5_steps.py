"""Per-step wall time of the 28x28 joint step from a cold start (is the warm-up long enough?)"""
import argparse, os, sys, time
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
prior = ProgramPrior(vocab, hidden_size=256).to(dev)
for p in prior.parameters():
    p.requires_grad_(False)
torch.manual_seed(5)
nmn = NeuralModuleNetwork(vocab, image_feature_size=(1024, 28, 28)).to(dev)
pg, qr = ProgramGenerator(vocab, max_decoding_steps=40).to(dev), QuestionReconstructor(vocab).to(dev)
batch = bench.device_batch(vocab, 128, 5000, dev, image_feature_size=(1024, 28, 28), deep=True)
bench.fit_program_generator(pg, vocab, batch, dev, 1500, 0.95)
trainer = JointTrainingStep(pg, qr, prior, nmn, **bench.JOINT)
ts = []
for i in range(40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    trainer.step(batch)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join("%.1f" % t for t in ts))
print(torch.cuda.memory_stats()["num_alloc_retries"], torch.cuda.memory_reserved() / 1e9)
# the same without a synchronisation per step (what bench.py times), with the host profiled
import cProfile, pstats
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for i in range(14):
    trainer.step(batch)
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("free-running: %.2f ms per step (host loop %.2f)" % ((t2 - t0) / 14 * 1e3, (t1 - t0) / 14 * 1e3))
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
