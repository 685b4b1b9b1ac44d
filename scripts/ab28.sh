#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; python -c "import torch" >/dev/null 2>&1
for i in 1 2; do
for cfg in "X=1" "PNMN_CONV_XCD_ROUNDROBIN=1" "PNMN_NO_WEIGHT_SORT=1" "PNMN_CONV_XCD_ROUNDROBIN=1 PNMN_NO_WEIGHT_SORT=1"; do
  env $cfg timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['joint_training_28x28']; r=s['roofline']['kernels']
print('%-60s' % '$cfg', 'headline %.2f' % d['ms_per_step'], '28x28 %.2f ms' % s['ms_per_step'], 'conv %.1f TF %.2f ms, wgrad %.1f TF %.2f ms' % (r['conv_nhwc']['tflops'], r['conv_nhwc']['ms_per_step'], r['conv_wgrad']['tflops'], r['conv_wgrad']['ms_per_step']))"
done; done
