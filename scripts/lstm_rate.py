"""Launch time of one LSTM layer (forward, forward + backward) by batch size."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd")]
import torch
from probnmn.modules.seq2seq_base import _LSTMLayerSeq
dev = torch.device("cuda:0")
T, H = 46, 256
for B in (64, 128, 256, 512, 1024):
    xp = (torch.randn(B, T, 4 * H, device=dev) * 0.5).requires_grad_(True)
    w = (torch.randn(4 * H, H, device=dev) * 0.05).requires_grad_(True)
    dhs = torch.randn(B, T, H, device=dev)
    def fwd():
        with torch.no_grad(): _LSTMLayerSeq.apply(xp, w)
    def both():
        _LSTMLayerSeq.apply(xp, w).backward(dhs)
    out = []
    for f in (fwd, both):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 100)
    print("B=%4d  fwd %.1f us (%.2f us/step)   fwd+bwd (with the weight-gradient GEMM) %.1f us" % (B, out[0], out[0] / T, out[1]))
