import os, sys, time, torch
sys.path.insert(0, "probnmn-clevr_amd")
from probnmn.modules.seq2seq_base import _LSTMLayerSeq
dev = torch.device("cuda:0")
for B in (128, 512, 1024):
    T, H = 46, 256
    xp = torch.randn(B, T, 4 * H, device=dev) * 0.5
    w = torch.randn(4 * H, H, device=dev) * 0.05
    dhs = torch.randn(B, T, H, device=dev)
    for mode in ("0", "1"):
        os.environ["PNMN_LSTM_CLUSTER"] = mode
        x = xp.clone().requires_grad_(True)
        for phase in ("fwd", "fwd+bwd"):
            def go():
                hs = _LSTMLayerSeq.apply(x, w)
                if phase != "fwd":
                    hs.backward(dhs)
            for _ in range(3): go()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): go()
            torch.cuda.synchronize()
            print("B=%d cluster=%s %s: %.3f ms" % (B, mode, phase, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
