#!/bin/bash
# which earlier side measurement slows bench.py's 28x28 side object down?
cd $GRAFT_REPO_ROOT
python -c "import torch" >/dev/null 2>&1
for S in joint_training_ingest,joint_training_28x28 joint_training_28x28; do
  timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 4 --warmup 2 --sides $S 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
j=d['joint_training_28x28']
print('%-60s 28x28 %.2f ms  busy %.2f blocked %.2f' % ('$S', j['ms_per_step'], j['host_busy_ms_per_step'], j['host_blocked_ms_per_step']))"
done | tee gpurun_out/r04_bisect28.txt
