#!/bin/bash
# usage: bash scripts/r03_ab_env2.sh TAG "BATCH ENV" ...
cd $GRAFT_REPO_ROOT
TAG=$1; shift
python -c "import torch" >/dev/null 2>&1
for rep in 1 2; do
for V in "$@"; do
  B=${V%% *}; E=${V#* }
  env $E timeout 300 python bench.py --batch $B --steps 60 --warmup 10 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$V', d['value'], d['ms_per_step'])" | tee -a gpurun_out/${TAG}_ab.txt
done
done
