#!/bin/bash
# merged conv segments: kernel tests, then A/B at 1024 (with the conv roofline) and at 128 questions
cd $GRAFT_REPO_ROOT
TAG=${1:-r04c}
python -c "import torch" >/dev/null 2>&1
timeout 1200 python -m pytest -x -q -m gpu tests/test_hip_kernels.py tests/test_nmn_gpu.py tests/test_trunk_planner.py 2>&1 | tail -2
for V in "PNMN_CONV_MERGED=1" "PNMN_CONV_MERGED=0" "PNMN_CONV_SEG_COST=0.1" "PNMN_CONV_MERGED=1" "PNMN_CONV_MERGED=0" "PNMN_CONV_SEG_COST=0.1" "PNMN_CONV_SEG_COST=0.0"; do
  env $V timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('$V', d['value'], d['ms_per_step'], 'conv_nhwc', r['achieved'], r['tflops_per_pass'], {k: (v['tflops'], v['ms_per_step']) for k, v in r['kernels']['conv_nhwc']['by_call_site'].items() if 'module' in k or 'stem conv1' in k})" | tee -a gpurun_out/${TAG}_ab.txt
done
bash scripts/r03_ab_env.sh ${TAG}_b128 128 "PNMN_CONV_MERGED=1" "PNMN_CONV_MERGED=0" "PNMN_CONV_SEG_COST=0.1"
