"""The NMN alone at the 128-question shard's 64 sampled rows: module-training steps (ground-truth programs: same launches as
the joint step's trunk, nothing beside them) -- how long is the trunk's own chain?  usage: python scripts/r06_trunk_alone.py [rows] [conv_cus]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd")]
import torch
import bench
from probnmn.models import NeuralModuleNetwork
from probnmn.trainers.module_training import ModuleTrainingStep
from probnmn.vocabulary import Vocabulary

dev = torch.device("cuda:0")
vocab = Vocabulary.clevr()
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cus = int(sys.argv[2]) if len(sys.argv) > 2 else 0
nmn = NeuralModuleNetwork(vocab).to(dev)
step = ModuleTrainingStep(nmn, lr=1e-4, weight_decay=0.0, report_metrics=False)
batch = bench.device_batch(vocab, B, 3000, dev)
batch["program"] = batch["program"].cpu()
nmn.engine.conv_cus = cus
for _ in range(10):
    nmn.engine.conv_cus = cus
    step.step(batch)
torch.cuda.synchronize()
marks = []
for phase in ("whole step",):
    t0 = time.perf_counter()
    for _ in range(100):
        nmn.engine.conv_cus = cus
        step.step(batch)
    torch.cuda.synchronize()
    print("module_training, %d rows, conv_cus %d: %.3f ms per step" % (B, cus, (time.perf_counter() - t0) * 10))
# forward only / backward only split with events
e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
fw = bw = 0.0
import probnmn.trainers.module_training as mt
for _ in range(50):
    nmn.engine.conv_cus = cus
    step.optimizer.zero_grad()
    e[0].record()
    out = nmn(batch["image"], batch["program"], batch["answer"])
    loss = out["loss"].mean()
    e[1].record()
    loss.backward()
    e[2].record()
    step.optimizer.step()
    e[3].record()
    torch.cuda.synchronize()
    fw += e[0].elapsed_time(e[1]); bw += e[1].elapsed_time(e[2])
print("  forward %.3f ms, backward %.3f ms (synchronised per step)" % (fw / 50, bw / 50))
