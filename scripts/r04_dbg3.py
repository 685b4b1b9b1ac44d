"""Debug aid: per golden program (batch of 2), gradients of the network with the conv kernel of this process
(PNMN_CONV_STREAM) saved to /tmp; a second run with the other kernel compares parameter by parameter."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from fixtures import VALIDITY_CASES, encode_programs
from probnmn.models.nmn import NeuralModuleNetwork
from probnmn.vocabulary import Vocabulary
mode = os.environ.get("PNMN_CONV_STREAM", "1")
vocab = Vocabulary.clevr(); torch.manual_seed(0)
net = NeuralModuleNetwork(vocab); stoi = vocab.get_token_to_index_vocabulary("programs")
dev = torch.device("cuda:0"); net.to(dev).train()
g = torch.Generator().manual_seed(1)
res = {}
for ci, case in enumerate(VALIDITY_CASES):
    programs = encode_programs([case, case], stoi)
    features = torch.relu(torch.randn(2, 1024, 14, 14, generator=g)); answers = torch.randint(0, 28, (2,), generator=g)
    net.zero_grad(set_to_none=True)
    out = net(features.to(dev), programs.to(dev), answers.to(dev))
    out["loss"].mean().backward()
    res[ci] = {n: (p.grad.detach().cpu().clone() if p.grad is not None else None) for n, p in net.named_parameters()}
    res[ci]["__loss"] = out["loss"].detach().cpu()
path = "/tmp/r04_grads_%s.pt" % mode
torch.save(res, path)
other = "/tmp/r04_grads_%s.pt" % ("0" if mode == "1" else "1")
if os.path.exists(other):
    ref = torch.load(other)
    for ci in res:
        worst, wname = 0.0, ""
        for n, gq in res[ci].items():
            if n == "__loss" or gq is None or ref[ci][n] is None: continue
            sc = float(ref[ci][n].abs().max())
            if sc == 0: continue
            e = float((gq - ref[ci][n]).abs().max()) / sc
            if e > worst: worst, wname = e, n
        print("case %2d  %-60s worst %.2e  %s  dloss %.1e" % (ci, " ".join(VALIDITY_CASES[ci])[:60], worst, wname, float((res[ci]["__loss"] - ref[ci]["__loss"]).abs().max())))
