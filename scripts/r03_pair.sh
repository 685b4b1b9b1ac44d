#!/bin/bash
cd $GRAFT_REPO_ROOT
TAG=${1:-r03p}
python -c "import torch" >/dev/null 2>&1
timeout 1200 python -m pytest -x -q -m gpu tests/test_seq2seq_gpu.py tests/test_joint_gpu.py tests/test_dp_trainers_gpu.py tests/test_nmn_gpu.py tests/test_full_size_gpu.py tests/test_seq2seq_overrides.py > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
tail -6 gpurun_out/${TAG}_pytest.log
for V in "PNMN_PAIR_DECODERS=1" "PNMN_PAIR_DECODERS=0" "PNMN_PAIR_DECODERS=1" "PNMN_PAIR_DECODERS=0"; do
  env $V timeout 300 python bench.py --batch 128 --steps 80 --warmup 10 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$V b128', d['value'], d['ms_per_step'], 'host busy', d['host_busy_ms_per_step'], 'blocked', d['host_blocked_ms_per_step'])" | tee -a gpurun_out/${TAG}_ab.txt
done
