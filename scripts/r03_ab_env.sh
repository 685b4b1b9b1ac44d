#!/bin/bash
# usage: bash scripts/r03_ab_env.sh TAG BATCH "ENV1" "ENV2" ...   (each env string is run twice, interleaved)
cd $GRAFT_REPO_ROOT
TAG=$1; B=$2; shift 2
python -c "import torch" >/dev/null 2>&1
for rep in 1 2; do
for V in "$@"; do
  env $V timeout 300 python bench.py --batch $B --steps 80 --warmup 10 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$V b$B', d['value'], d['ms_per_step'], 'host busy', d['host_busy_ms_per_step'], 'blocked', d['host_blocked_ms_per_step'])" | tee -a gpurun_out/${TAG}_ab.txt
done
done
