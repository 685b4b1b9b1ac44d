"""Module weight-gradient launch time against the items per job (wgrad_chunk): how much of it is the atomic flush?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd")]
import torch
import bench
from probnmn.models import NeuralModuleNetwork
from probnmn.trainers.module_training import ModuleTrainingStep
from probnmn.vocabulary import Vocabulary

B = 1024
dev = torch.device("cuda:0")
vocab = Vocabulary.clevr()
batch = bench.device_batch(vocab, B, 1000, dev)
batch["program"] = batch["program"].cpu()
for chunk in (4, 8, 16, 32, 64):
    torch.manual_seed(0)
    nmn = NeuralModuleNetwork(vocab).to(dev)
    nmn.engine.ensure_arena()
    nmn.engine.planner_config.wgrad_chunk = chunk
    step = ModuleTrainingStep(nmn, lr=1e-4, report_metrics=False)
    for _ in range(3):
        step.step(batch)
    torch.cuda.synchronize()
    best = {}
    for _ in range(3):
        nmn.engine.begin_trace()
        step.step(batch)
        torch.cuda.synchronize()
        for kern, what, flops, ms, _, _ in nmn.engine.end_trace():
            if kern == "conv_wgrad":
                best[what] = min(best.get(what, 1e9), ms)
                best[what + " TF"] = flops / best[what] / 1e9
    print("chunk %2d: " % chunk + "  ".join("%s %.3f ms (%.0f TF)" % (w, best[w], best[w + " TF"]) for w in ("module wgrad", "stem conv2 wgrad", "classifier wgrad", "stem conv1 wgrad")))
    step.close() if hasattr(step, "close") else None
    del step, nmn
