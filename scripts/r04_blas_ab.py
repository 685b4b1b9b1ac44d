"""Joint step time with torch's GEMMs on rocBLAS vs hipBLASLt (host cost per call differs): python r04_blas_ab.py <B> <backend>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd")]
import torch
B, backend = int(sys.argv[1]), sys.argv[2]
if backend != "default":
    torch.backends.cuda.preferred_blas_library(backend)
print("preferred:", torch.backends.cuda.preferred_blas_library())
import bench
from probnmn.models import NeuralModuleNetwork, ProgramGenerator, ProgramPrior, QuestionReconstructor
from probnmn.trainers.joint_training import JointTrainingStep
from probnmn.vocabulary import Vocabulary
dev = torch.device("cuda:0")
vocab = Vocabulary.clevr()
torch.manual_seed(0)
nmn = NeuralModuleNetwork(vocab).to(dev)
pg, qr, prior = ProgramGenerator(vocab).to(dev), QuestionReconstructor(vocab).to(dev), ProgramPrior(vocab, hidden_size=256).to(dev)
batch = bench.device_batch(vocab, B, 1000, dev)
bench.fit_program_generator(pg, vocab, batch, dev, 1500, 0.95)
step = JointTrainingStep(pg, qr, prior, nmn, **bench.JOINT)
for _ in range(12): step.step(batch)
torch.cuda.synchronize()
res = []
for rep in range(3):
    N = 40 if B <= 256 else 15
    t0 = time.perf_counter()
    for _ in range(N): step.step(batch)
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / N * 1e3)
print("B=%d %s: %s ms/step" % (B, backend, " ".join("%.3f" % r for r in res)))
