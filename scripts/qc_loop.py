"""N question_coding steps at 512 questions (BASELINE configs[2]) -- for `rocprofv3 --kernel-trace` +
`profiles/summarize.py --steady`.  usage: python scripts/qc_loop.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd")]
import torch
import bench
from probnmn.models import ProgramGenerator, ProgramPrior, QuestionReconstructor
from probnmn.trainers.joint_training import QuestionCodingStep
from probnmn.vocabulary import Vocabulary

dev = torch.device("cuda:0")
vocab = Vocabulary.clevr()
torch.manual_seed(0)
pg, qr, prior = ProgramGenerator(vocab).to(dev), QuestionReconstructor(vocab).to(dev), ProgramPrior(vocab, hidden_size=256).to(dev)
batch = bench.device_batch(vocab, 512, 3000, dev)
bench.fit_program_generator(pg, vocab, batch, dev, 1500, 0.95)
step = QuestionCodingStep(pg, qr, prior, objective="ours", alpha=100.0, beta=0.1, delta=0.99, lr=1e-3)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for _ in range(8): step.step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N): step.step(batch)
torch.cuda.synchronize()
print("question_coding, 512 questions: %.2f ms per step" % ((time.perf_counter() - t0) / N * 1e3))
