"""Per-launch table of the conv kernels in one module_training step: items, work-groups, rounds of
256 CUs, measured time, TFLOP/s -- to see where the grouped launches lose efficiency."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd")]
import torch
import bench
from probnmn.models import NeuralModuleNetwork
from probnmn.trainers.module_training import ModuleTrainingStep
from probnmn.vocabulary import Vocabulary

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
vocab = Vocabulary.clevr()
torch.manual_seed(0)
nmn = NeuralModuleNetwork(vocab).to(dev)
batch = bench.device_batch(vocab, B, 1000, dev)
batch["program"] = batch["program"].cpu()
step = ModuleTrainingStep(nmn, lr=1e-4, report_metrics=False)
for _ in range(3):
    step.step(batch)
torch.cuda.synchronize()
nmn.engine.begin_trace()
step.step(batch)
torch.cuda.synchronize()
rows = []
for kern, what, flops, ms, nbytes, _ in nmn.engine.end_trace():
    rows.append((kern, what, flops, ms, nbytes))
tot = {}
print("%-11s %-22s %9s %9s %8s %10s" % ("kernel", "site", "GFLOP", "ms", "TF", "alg MB"))
for kern, what, flops, ms, nbytes in rows:
    print("%-11s %-22s %9.2f %9.4f %8.1f %10.1f" % (kern, what, flops / 1e9, ms, flops / ms / 1e9, nbytes / 1e6))
    t = tot.setdefault(kern, [0.0, 0.0]); t[0] += flops; t[1] += ms
for k, (f, ms) in tot.items():
    print("TOTAL %-11s %9.2f GFLOP %9.3f ms %8.1f TF" % (k, f / 1e9, ms, f / ms / 1e9))
