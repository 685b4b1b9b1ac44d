"""How far are the device's NMN gradients from the oracle's, against the oracle's own sensitivity to weight noise?
Per example (batch of 1, ground-truth program): relative error of every gradient tensor (max |diff| / max |ref|) for the
device and for oracles whose weights are perturbed by 1e-6 / 1e-5 / 1e-4 -- the number of tensors above 1e-3 tells a
flipped gate (few tensors, large) from rounding (all tensors, tiny)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch

from oracle import nmn_oracle
from probnmn.data.synthetic import synthetic_batch
from probnmn.models.nmn import NeuralModuleNetwork
from probnmn.vocabulary import Vocabulary

torch.set_num_threads(16)
dev = torch.device("cuda:0")
vocab = Vocabulary.clevr()
itos = vocab.get_index_to_token_vocabulary("programs")
torch.manual_seed(0)
net = NeuralModuleNetwork(vocab)
sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
net.to(dev).train()


def oracle_grads(state, batch):
    params = {k: v.detach().clone().requires_grad_(True) for k, v in state.items()}
    out = nmn_oracle.nmn_forward(params, itos, batch["image"], batch["program"], batch["answer"])
    out["loss"].mean().backward()
    return {k: p.grad for k, p in params.items() if p.grad is not None}, out["loss"].detach()


def errors(a, b):
    return {k: float((a[k] - b[k]).abs().max()) / (float(b[k].abs().max()) + 1e-20) for k in b if k in a and float(b[k].abs().max()) > 0}


def perturbed(state, seed, scale):
    g = torch.Generator().manual_seed(seed)
    return {k: v * (1.0 + scale * torch.randn(v.shape, generator=g)) for k, v in state.items()}


n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
big = synthetic_batch(vocab, n, seed=1000)
print("example: tensors with gradient | device: median, #>1e-3, worst | control 1e-6 | 1e-5 | 1e-4")
for i in range(n):
    b = {k: v[i:i + 1] for k, v in big.items()}
    ref, ref_loss = oracle_grads(sd, b)
    net.zero_grad(set_to_none=True)
    out = net(b["image"].to(dev), b["program"].to(dev), b["answer"].to(dev))
    out["loss"].mean().backward()
    torch.cuda.synchronize()
    got = {k: p.grad.detach().cpu() for k, p in net.named_parameters() if p.grad is not None}
    row = []
    for name, g in [("device", got)] + [("%.0e" % s, oracle_grads(perturbed(sd, 7, s), b)[0]) for s in (1e-6, 1e-5, 1e-4)]:
        e = errors(g, ref)
        v = np.array(sorted(e.values()))
        worst = max(e, key=e.get)
        row.append("%.1e %2d %.1e (%s)" % (np.median(v), int((v > 1e-3).sum()), v[-1], worst[-28:]))
    print("%2d: %3d | %s | loss diff %.1e" % (i, len(ref), " | ".join(row), float((out["loss"].detach().cpu() - ref_loss).abs().max())), flush=True)
