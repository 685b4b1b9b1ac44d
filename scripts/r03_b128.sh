#!/bin/bash
# b128 step: parity tests of the step, then timelines without a profiler (free-running and one sync per step), then A/B bench lines.
cd $GRAFT_REPO_ROOT
TAG=${1:-r03e}
python -c "import torch" >/dev/null 2>&1
timeout 900 python -m pytest -x -q -m gpu tests/test_joint_gpu.py tests/test_elbo_gpu.py tests/test_dp_trainers_gpu.py tests/test_full_size_gpu.py > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
tail -4 gpurun_out/${TAG}_pytest.log
timeout 300 python scripts/step_timeline.py 128 40 --free > gpurun_out/${TAG}_timeline_free.txt 2>&1
timeout 300 python scripts/step_timeline.py 128 40 > gpurun_out/${TAG}_timeline_sync.txt 2>&1
for V in "" "PNMN_FUSED_OBJECTIVE=0" "PNMN_NATIVE_PLANNER=0"; do
  env $V timeout 300 python bench.py --batch 128 --steps 60 --warmup 10 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$V', d['value'], d['ms_per_step'], 'host busy', d['host_busy_ms_per_step'], 'blocked', d['host_blocked_ms_per_step'])" | tee -a gpurun_out/${TAG}_ab.txt
done
cat gpurun_out/${TAG}_timeline_free.txt | tail -45
