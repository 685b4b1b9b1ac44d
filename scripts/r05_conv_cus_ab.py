"""A/B of the 128-question joint step: CUs the trunk's conv launches are cut for while the seq2seq kernels share the chip
(JointTrainingStep.shared_conv_cus), with the three decoders' backward grouped -- same process, alternating."""
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


def run(n=60):
    for _ in range(8): step.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): step.step(batch)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(3):
    for cus in (0, 128, 160, 192, 224, 256):
        step.shared_conv_cus = cus
        print("rep %d  shared_conv_cus %3d  %.3f ms" % (rep, cus, run()), flush=True)
