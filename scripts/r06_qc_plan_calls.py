import os, sys
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd")]
import torch
from probnmn import _hip
from probnmn.data.synthetic import synthetic_batch
from probnmn.models import ProgramGenerator, ProgramPrior, QuestionReconstructor
from probnmn.trainers.joint_training import QuestionCodingStep
from probnmn.vocabulary import Vocabulary
dev = torch.device("cuda:0"); vocab = Vocabulary.clevr(); torch.manual_seed(0)
models = [ProgramGenerator(vocab).to(dev), QuestionReconstructor(vocab).to(dev), ProgramPrior(vocab, hidden_size=256).to(dev)]
batch = synthetic_batch(vocab, 512, seed=1, with_image=False)
batch["supervision"][:] = 0; batch["supervision"][:256] = 1
db = {k: v.to(dev) for k, v in batch.items()}; db["supervision"] = batch["supervision"]
step = QuestionCodingStep(*models, objective="ours", alpha=100.0, beta=0.1, delta=0.99, lr=1e-4)
step.step(db); torch.cuda.synchronize()
plan = [p for p in step._plans.values() if p][0]
print("n", plan.n, "m", plan.m)
for lname in ("fwd_pg_enc", "fwd_qr", "fwd_prior"):
    print(lname, [c[2] for c in getattr(plan, lname)])
import numpy as np
jobs = np.zeros(2, _hip.LSTM_STACK_JOB); jobs["B"] = 256; jobs["T"] = 28; jobs[0]["dep"] = -1; jobs[1]["dep"] = 0
print("ws bytes for 256 rows:", _hip.lib().pnmn_lstm_stack_workspace_bytes(jobs.ctypes.data, 2, 0))
