#!/bin/bash
# b128: A/B of the stem placement, then the host's cProfile
cd $GRAFT_REPO_ROOT
TAG=${1:-r03f}
python -c "import torch" >/dev/null 2>&1
for V in "PNMN_X=0" "PNMN_STEM_AFTER_ENCODE=0" "PNMN_X=1" "PNMN_STEM_AFTER_ENCODE=0"; do
  env $V timeout 300 python bench.py --batch 128 --steps 80 --warmup 10 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$V', d['value'], d['ms_per_step'], 'host busy', d['host_busy_ms_per_step'], 'blocked', d['host_blocked_ms_per_step'])" | tee -a gpurun_out/${TAG}_ab.txt
done
timeout 300 python scripts/host_cprofile.py 128 > gpurun_out/${TAG}_host_cprofile.txt 2>&1
timeout 300 python scripts/step_timeline.py 128 40 --free > gpurun_out/${TAG}_timeline_free.txt 2>&1
head -60 gpurun_out/${TAG}_host_cprofile.txt
