#!/usr/bin/env python3
"""Side measurements (not the headline bench): questions/s of the question_coding and joint_training
iterations on one MI355X with synthetic CLEVR-shaped batches.

    python scripts/bench_phases.py --phase question_coding --batch 512
    python scripts/bench_phases.py --phase joint_training --batch 128
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "probnmn-clevr_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--phase", choices=["question_coding", "joint_training"], required=True)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()

    from probnmn.data.synthetic import synthetic_batch
    from probnmn.models import NeuralModuleNetwork, ProgramGenerator, ProgramPrior, QuestionReconstructor
    from probnmn.trainers.joint_training import JointTrainingStep, QuestionCodingStep
    from probnmn.vocabulary import Vocabulary

    dev = torch.device("cuda:0")
    vocab = Vocabulary.clevr()
    torch.manual_seed(0)
    pg, qr, prior = ProgramGenerator(vocab).to(dev), QuestionReconstructor(vocab).to(dev), ProgramPrior(vocab, hidden_size=256).to(dev)
    with_image = args.phase == "joint_training"
    batch = synthetic_batch(vocab, args.batch, seed=0, with_image=with_image)
    host_sup = batch["supervision"]
    batch = {k: v.to(dev) for k, v in batch.items()}
    batch["supervision"] = host_sup
    if args.phase == "question_coding":
        step = QuestionCodingStep(pg, qr, prior, lr=1e-3)
    else:
        nmn = NeuralModuleNetwork(vocab).to(dev)
        step = JointTrainingStep(pg, qr, prior, nmn, lr=1e-6)
    for _ in range(args.warmup):
        step.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step.step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"phase": args.phase, "batch": args.batch, "ms_per_step": round(dt * 1e3, 2),
                      "questions_per_s": round(args.batch / dt, 1)}))


if __name__ == "__main__":
    main()
