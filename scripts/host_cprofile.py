"""cProfile of the host side of joint-training steps (main thread only: the backward's launches are issued by the
autograd thread and show up as time inside loss.backward)."""
import cProfile, os, pstats, sys
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
for _ in range(8): step.step(batch)
torch.cuda.synchronize()
N = 40
pr = cProfile.Profile()
pr.enable()
for _ in range(N): step.step(batch)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
print("per step (ms): tottime / cumtime, %d steps" % N)
rows = []
for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
    rows.append((tt / N * 1e3, ct / N * 1e3, nc / N, "%s:%d(%s)" % (os.path.basename(fn), line, name)))
print("--- by tottime")
for r in sorted(rows, key=lambda r: -r[0])[:45]: print("%8.3f %8.3f %7.1f  %s" % r)
print("--- by cumtime")
for r in sorted(rows, key=lambda r: -r[1])[:45]: print("%8.3f %8.3f %7.1f  %s" % r)
