#!/bin/bash
cd $GRAFT_REPO_ROOT
TAG=${1:-r03u}
python -c "import torch" >/dev/null 2>&1
timeout 1200 python -m pytest -x -q -m gpu tests/test_seq2seq_gpu.py tests/test_joint_gpu.py tests/test_dp_trainers_gpu.py tests/test_full_size_gpu.py > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
tail -5 gpurun_out/${TAG}_pytest.log
for rep in 1 2; do for V in "PNMN_PAIR_DECODERS=1" "PNMN_PAIR_DECODERS=0"; do
  env $V timeout 300 python bench.py --batch 128 --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$V b128', d['value'], d['ms_per_step'])" | tee -a gpurun_out/${TAG}_ab.txt
  env $V timeout 400 python - <<'P' 2>/dev/null | tee -a gpurun_out/${TAG}_ab.txt
import os, sys, time, torch
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "probnmn-clevr_amd")]
import bench
from probnmn.models import ProgramGenerator, ProgramPrior, QuestionReconstructor
from probnmn.trainers.joint_training import QuestionCodingStep
from probnmn.vocabulary import Vocabulary
dev = torch.device("cuda:0"); vocab = Vocabulary.clevr(); torch.manual_seed(0)
pg, qr, prior = ProgramGenerator(vocab).to(dev), QuestionReconstructor(vocab).to(dev), ProgramPrior(vocab, hidden_size=256).to(dev)
for B in (512, 256):
    b = bench.device_batch(vocab, B, 3000, dev)
    step = QuestionCodingStep(pg, qr, prior, objective="ours", alpha=100.0, beta=0.1, delta=0.99, lr=1e-3)
    e, h, bl = bench.timed(lambda: step.step(b), 30, 10, dev, 1, step)
    print(os.environ.get("PNMN_PAIR_DECODERS"), "question_coding", B, "%.3f ms" % (e / 30 * 1e3), "%.0f q/s" % (B * 30 / e))
P
done; done
