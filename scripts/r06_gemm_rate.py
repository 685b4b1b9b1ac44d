"""pnmn_gemm rate on square and seq2seq / FC shapes against torch (hipBLASLt).  usage: python scripts/r06_gemm_rate.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd")]
import numpy as np
import torch
from probnmn import _hip

dev = torch.device("cuda:0")
lib, st = _hip.lib(), _hip.stream_ptr(dev)


def run(name, M, N, K, ta, tb, split):
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    d = np.zeros(1, _hip.GEMM_DESC)
    d["a"], d["b"], d["c"] = A.data_ptr(), B.data_ptr(), C.data_ptr()
    d["lda"], d["ldb"], d["ldc"] = A.stride(0), B.stride(0), C.stride(0)
    d["M"], d["N"], d["K"], d["flags"], d["split_k"] = M, N, K, ta * 1 + tb * 2, split
    ws = torch.zeros(max(int(lib.pnmn_gemm_workspace_bytes(M, N, split)), 4), dtype=torch.uint8, device=dev)
    d["workspace"] = ws.data_ptr()
    for _ in range(3):
        lib.pnmn_gemm(d.ctypes.data, 1, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.pnmn_gemm(d.ctypes.data, 1, st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    a, b = (A.t() if ta else A), (B.t() if tb else B)
    for _ in range(3):
        ref = a @ b
    e0.record()
    for _ in range(20):
        ref = a @ b
    e1.record()
    torch.cuda.synchronize()
    mt = e0.elapsed_time(e1) / 20
    err = float((C - ref).abs().max()) / float(ref.abs().max())
    print("%-14s M %6d N %6d K %6d ta %d tb %d split %2d: %8.1f us %6.1f TF | torch %8.1f us %6.1f TF | err %.1e"
          % (name, M, N, K, ta, tb, split, ms * 1e3, 2.0 * M * N * K / ms / 1e9, mt * 1e3, 2.0 * M * N * K / mt / 1e9, err))


for ta, tb in ((0, 1), (0, 0), (1, 0)):
    run("square", 4096, 4096, 4096, ta, tb, 1)
run("xp2 b1024", 47104, 1024, 256, 0, 1, 1)
run("dx b1024", 47104, 256, 1024, 0, 0, 1)
run("wgrad b1024", 1024, 256, 47104, 1, 0, 32)
run("fc1 fwd 512", 512, 1024, 50176, 0, 1, 12)
run("fc1 dx 512", 512, 50176, 1024, 0, 0, 1)
run("fc1 dw 512", 1024, 50176, 512, 1, 0, 1)
run("fc1 fwd 64", 64, 1024, 50176, 0, 1, 48)
run("fc1 dx 64", 64, 50176, 1024, 0, 0, 1)
run("fc1 dw 64", 1024, 50176, 64, 1, 0, 1)
