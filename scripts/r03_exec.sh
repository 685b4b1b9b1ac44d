#!/bin/bash
# trunk executor: A/B against the grouped launches at 1024 and 128 questions (every run under its own timeout)
cd $GRAFT_REPO_ROOT
TAG=${1:-r04e}
python -c "import torch" >/dev/null 2>&1
timeout 300 python -m pytest -x -q -m gpu tests/test_nmn_gpu.py -k "executor" 2>&1 | tail -3
for V in "PNMN_TRUNK_EXEC=0" "PNMN_TRUNK_EXEC=1" "PNMN_TRUNK_EXEC=0" "PNMN_TRUNK_EXEC=1"; do
  env $V timeout 150 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$V', d['value'], d['ms_per_step'])" | tee -a gpurun_out/${TAG}_ab.txt
done
for rep in 1 2; do
for V in "PNMN_TRUNK_EXEC=0" "PNMN_TRUNK_EXEC=1"; do
  env $V timeout 150 python bench.py --batch 128 --steps 80 --warmup 10 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$V b128', d['value'], d['ms_per_step'], 'host busy', d['host_busy_ms_per_step'], 'blocked', d['host_blocked_ms_per_step'])" | tee -a gpurun_out/${TAG}_b128_ab.txt
done
done
