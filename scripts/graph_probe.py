"""Experiment (round 2, rejected): capture a teacher-forced seq2seq pass (forward + backward) in hipGraphs via
torch.cuda.make_graphed_callables and compare with the eager pass (values, gradients, host time).

Findings: (1) the capture is bit-exact once the multi-CU kernels' hand-off counters are zeroed by a kernel
instead of hipMemsetAsync (a memset NODE ran out of order with the kernel nodes around it and trapped a
running multi-CU kernel; csrc/cluster.h); (2) the backward capture segfaults in capture_end if ANY eager
autograd graph over the same parameters is alive (AccumulateGrad nodes bound to the default stream), so
captures must happen at the very start of an iteration; (3) host time of one pass drops 2.5 -> 0.7 ms, but
with the three teacher-forced passes of the joint step captured as one forward + one backward graph the
128-question step got SLOWER (9.95 -> 10.3 ms; 1024 questions: 35.9 -> 36.7 ms): hipGraph replay on this
stack costs about as much host and GPU time as it saves.  Not shipped."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd")]
import torch
from torch import nn
from probnmn.models import QuestionReconstructor
from probnmn.vocabulary import Vocabulary

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
vocab = Vocabulary.clevr()
torch.manual_seed(0)
qr = QuestionReconstructor(vocab).to(dev)
qr.train()

class TF(nn.Module):
    def __init__(self, m):
        super().__init__()
        self.m = m
    def forward(self, src, tgt):
        return self.m(src, tgt, "sampling", False)["loss"]

g = torch.Generator().manual_seed(1)
def toks(T, V):
    out = torch.zeros(B, T, dtype=torch.long)
    lens = torch.randint(3, T + 1, (B,), generator=g)
    for i in range(B):
        out[i, : lens[i]] = torch.randint(4, V, (int(lens[i]),), generator=g)
    return out.to(dev)
src, tgt = toks(26, 44), toks(45, 100)
src2, tgt2 = toks(26, 44), toks(45, 100)
eager = TF(qr)
w = torch.randn(B, device=dev)

def run(mod, s, t):
    for p in qr.parameters(): p.grad = None
    loss = mod(s, t)
    (loss * w).sum().backward()
    return loss.detach().clone(), [p.grad.detach().clone() for p in qr.parameters()]

ref = run(eager, src2, tgt2)
graphed = torch.cuda.make_graphed_callables(TF(qr), (src, tgt), allow_unused_input=True)
got = run(graphed, src2, tgt2)
print("loss max diff", float((ref[0] - got[0]).abs().max()))
print("grad max rel diff", max(float((a - b).abs().max() / (a.abs().max() + 1e-12)) for a, b in zip(ref[1], got[1])))
for name, mod in (("eager", eager), ("graphed", graphed)):
    for _ in range(3): run(mod, src, tgt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): run(mod, src, tgt)
    host = (time.perf_counter() - t0) / 20
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / 20
    print("%-8s rows=%d: host %.2f ms per fwd+bwd, wall %.2f ms" % (name, B, host * 1e3, total * 1e3))
