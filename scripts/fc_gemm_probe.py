"""Time of the NMN's fully connected layer (classifier.4: 50176 -> 1024, fp32) at the headline's 512 sampled
rows: forward, data gradient, weight gradient through torch / hipBLASLt, and a hand-split-K formulation."""
import os, sys, time
import torch
import torch.nn.functional as F
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K, N = 50176, 1024
x = torch.randn(B, K, device=dev)
W = torch.randn(N, K, device=dev) * 0.01
dy = torch.randn(B, N, device=dev)

def clock(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

flops = 2.0 * B * K * N
for name, fn in (("fwd  x @ W^T", lambda: F.linear(x, W)),
                 ("dgrad dy @ W", lambda: dy @ W),
                 ("wgrad dy^T @ x", lambda: dy.t() @ x)):
    ms = clock(fn)
    print("%-16s %.3f ms  %.1f TFLOP/s" % (name, ms, flops / ms / 1e9))
for S in (4, 8, 16, 32):
    xs = x.view(B, S, K // S).transpose(0, 1)            # [S, B, K/S]
    Ws = W.view(N, S, K // S).permute(1, 2, 0)           # [S, K/S, N]
    ms = clock(lambda: torch.bmm(xs, Ws).sum(0))
    print("fwd split-K %2d (bmm + sum) %.3f ms  %.1f TFLOP/s" % (S, ms, flops / ms / 1e9))
# wgrad alternatives: output [N, K] is large (tiles plentiful); dgrad output [B, K] too
ms = clock(lambda: torch.mm(dy.t().contiguous(), x)); print("wgrad with contiguous dy^T %.3f ms %.1f TF" % (ms, flops / ms / 1e9))
ms = clock(lambda: (x.t() @ dy).t()); print("wgrad as (x^T @ dy)^T %.3f ms %.1f TF" % (ms, flops / ms / 1e9))
