"""Library GEMM calls of a joint-training step by call site: aten::mm / addmm / bmm / baddbmm grouped by
operand shapes, with device time, FLOPs and TFLOP/s.   usage: python scripts/r05_gemm_sites.py [questions]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd")]
import torch
from torch.profiler import profile, ProfilerActivity
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
os.environ["PNMN_NMN_STREAM"] = "0"  # one stream: a kernel's duration is its own
step = JointTrainingStep(pg, qr, prior, nmn, **bench.JOINT)
for _ in range(10): step.step(batch)
torch.cuda.synchronize()
N = 10
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(N): step.step(batch)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key not in ("aten::mm", "aten::addmm", "aten::bmm", "aten::baddbmm"):
        continue
    shapes = [s for s in e.input_shapes if s]
    try:
        if e.key == "aten::mm":
            (m, k), (_, n) = shapes[0], shapes[1]
            flops = 2.0 * m * k * n
        elif e.key == "aten::addmm":
            (m, k), (_, n) = shapes[1], shapes[2]
            flops = 2.0 * m * k * n
        elif e.key == "aten::bmm":
            (b, m, k), (_, _, n) = shapes[0], shapes[1]
            flops = 2.0 * b * m * k * n
        else:
            (b, m, k), (_, _, n) = shapes[1], shapes[2]
            flops = 2.0 * b * m * k * n
    except Exception:
        flops = 0.0
    dev_us = getattr(e, "device_time_total", None)
    if dev_us is None:
        dev_us = e.cuda_time_total
    rows.append((dev_us / N, e.count / N, e.key, str(shapes), flops, e.self_cpu_time_total / N))
rows.sort(reverse=True)
print("%d questions, per step: device us | calls | op | shapes | GFLOP per call | TFLOP/s | host us" % B)
tot = 0.0
for us, calls, key, shapes, flops, host in rows:
    tot += us
    print("%9.1f %5.1f  %-12s %-44s %8.3f %7.1f %8.1f" % (us, calls, key, shapes, flops / 1e9, flops * calls / max(us, 1e-9) / 1e6, host))
print("total %.1f us of library GEMMs per step" % tot)
