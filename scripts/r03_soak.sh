#!/bin/bash
# soak of the trunk executor: the equivalence tests and the 128 / 1024-question steps with it, repeatedly, each under a timeout
cd $GRAFT_REPO_ROOT
python -c "import torch" >/dev/null 2>&1
export PNMN_TRUNK_EXEC=1
for i in 1 2 3 4 5 6; do
  timeout 120 python -m pytest -q -m gpu tests/test_nmn_gpu.py -k "executor" 2>&1 | tail -1
  timeout 120 python bench.py --batch 128 --steps 300 --warmup 10 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b128 exec', d['ms_per_step'])" || echo "b128 run $i FAILED rc=$?"
  timeout 120 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b1024 exec', d['ms_per_step'])" || echo "b1024 run $i FAILED rc=$?"
done
