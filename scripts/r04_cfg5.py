"""The 28x28 side object of bench.py on its own (A/B aid): python scripts/r04_cfg5.py"""
import argparse, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd")]
import torch
import bench
from probnmn.models import ProgramPrior
from probnmn.vocabulary import Vocabulary

dev = torch.device("cuda:0")
vocab = Vocabulary.clevr()
torch.manual_seed(0)
prior = ProgramPrior(vocab, hidden_size=256).to(dev)
for p in prior.parameters():
    p.requires_grad_(False)
args = argparse.Namespace(batch28=128, fit_iters=1500, fit_target=0.95)
out = bench.config5_side(vocab, prior, dev, 0, 1, args)
r = out["roofline"]
print("28x28: %.2f ms  host busy %.2f blocked %.2f  single-stream %.2f  conv %.3f" % (
    out["ms_per_step"], out["host_busy_ms_per_step"], out["host_blocked_ms_per_step"], r["single_stream_step_ms"], r["frac"]))
