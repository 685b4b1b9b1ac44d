"""The 28x28 / 40-token side measurement of bench.py alone (A/B of schedule switches through the environment)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
bench.setup_paths() if hasattr(bench, "setup_paths") else None
from probnmn.models import ProgramPrior
from probnmn.vocabulary import Vocabulary

dev = torch.device("cuda:0")
vocab = Vocabulary.clevr()
torch.manual_seed(0)
prior = ProgramPrior(vocab, hidden_size=256).to(dev)
args = argparse.Namespace(batch28=int(os.environ.get("C5_BATCH", "128")), fit_iters=1500, fit_target=0.95)
out = bench.config5_side(vocab, prior, dev, 0, 1, args)
print("c5", os.environ.get("TAGV", ""), out["value"], out["ms_per_step"], "conv_nhwc", out["roofline"]["achieved"], flush=True)
