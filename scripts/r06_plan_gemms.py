"""Every pnmn_gemm launch of the seq2seq plan at a given batch size, timed alone (events, 20 repetitions), with the problems of
the launch and what torch (hipBLASLt) takes for the same products one by one.
usage: python scripts/r06_plan_gemms.py [questions=1024]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd")]
import numpy as np
import torch
from probnmn import _hip
from probnmn.data.synthetic import synthetic_batch
from probnmn.models import ProgramGenerator, ProgramPrior, QuestionReconstructor
from probnmn.trainers.joint_training import QuestionCodingStep
from probnmn.vocabulary import Vocabulary

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
vocab = Vocabulary.clevr()
torch.manual_seed(0)
models = [ProgramGenerator(vocab), QuestionReconstructor(vocab), ProgramPrior(vocab, hidden_size=256)]
for m in models:
    m.to(dev)
batch = synthetic_batch(vocab, B, seed=1, with_image=False)
batch["supervision"][:] = 0
batch["supervision"][:B // 2] = 1
dbatch = {k: v.to(dev) for k, v in batch.items()}
dbatch["supervision"] = batch["supervision"]
step = QuestionCodingStep(*models, objective="ours", alpha=100.0, beta=0.1, delta=0.99, lr=1e-4)
for _ in range(3):
    step.step(dbatch)
torch.cuda.synchronize()
plan = [p for p in step._plans.values() if p is not False][0]
recs = {r.ctypes.data: r for r in plan._keep if isinstance(r, np.ndarray) and r.dtype == _hip.GEMM_DESC}


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


total_us = total_t = total_gf = 0.0
for lname in ("fwd_pg_enc", "fwd_pg", "fwd_pg_finish", "fwd_qr", "fwd_prior", "bwd_a", "aux_a", "bwd_b", "aux_b"):
    for fn, args, name in getattr(plan, lname, []):
        if name != "pnmn_gemm_cus":
            continue
        rec = recs[args[0]]
        us = timed(lambda: fn(*args))
        gf, tus, lines = 0.0, 0.0, []
        for r in rec:
            M, N, K = int(r["M"]), int(r["N"]), int(r["K"])
            ta, tb = bool(r["flags"] & 1), bool(r["flags"] & 2)
            g = 2.0 * M * N * K / 1e9
            gf += g
            a = torch.randn((K, M) if ta else (M, K), device=dev)
            b = torch.randn((N, K) if tb else (K, N), device=dev)
            a, b = (a.t() if ta else a), (b.t() if tb else b)
            t = timed(lambda: a @ b, 10)
            tus += t
            lines.append("      M %6d N %5d K %6d ta %d tb %d split %2d  %6.2f GF   torch %7.1f us %6.1f TF"
                         % (M, N, K, ta, tb, int(r["split_k"]), g, t, g / t * 1e3))
        print("%-14s %d problems %7.2f GF %8.1f us %6.1f TF   (torch one by one %8.1f us %6.1f TF)" % (lname, len(rec), gf, us, gf / us * 1e3, tus, gf / tus * 1e3))
        print("\n".join(lines))
        total_us, total_t, total_gf = total_us + us, total_t + tus, total_gf + gf
print("all launches: %.1f GF, %.1f us = %.1f TF; torch %.1f us = %.1f TF" % (total_gf, total_us, total_gf / total_us * 1e3, total_t, total_gf / total_t * 1e3))
