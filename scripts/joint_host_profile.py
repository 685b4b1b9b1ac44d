"""Where does the host spend a joint_training step?  (time blocked waiting for the GPU vs. busy)"""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
batch = bench.device_batch(vocab, B, 1000, dev)
bench.fit_program_generator(pg, vocab, batch, dev, 1500, 0.95)
step = JointTrainingStep(pg, qr, prior, nmn, **bench.JOINT)
waits = []
orig = torch.cuda.Event.synchronize
def timed_sync(self):
    t0 = time.perf_counter(); orig(self); waits.append(time.perf_counter() - t0)
torch.cuda.Event.synchronize = timed_sync
for _ in range(3): step.step(batch)
torch.cuda.synchronize(); waits.clear()
t0 = time.perf_counter()
N = 10
for _ in range(N): step.step(batch)
host = time.perf_counter() - t0
torch.cuda.synchronize()
total = time.perf_counter() - t0
print("B=%d: %.2f ms/step total, host loop %.2f ms/step, of which blocked in event waits %.2f ms/step (%d waits)" % (
    B, total / N * 1e3, host / N * 1e3, sum(waits) / N * 1e3, len(waits)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step.step(batch)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumulative").print_stats("probnmn|autograd|bench", 45)
