#!/bin/bash
# A/B of the 1x1 convolutions inside the step: old kernel (PNMN_CONV_STREAM=1) vs streamed (2)
cd $GRAFT_REPO_ROOT
python -c "import torch" >/dev/null 2>&1
for s in 1 2 1 2; do
  PNMN_CONV_STREAM=$s timeout 900 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
r=d['roofline']
print('stream=$s  %.2f ms  %.0f q/s   conv_nhwc %.1f TF frac %.3f' % (d['ms_per_step'], d['value'], r['achieved'], r['frac']), r.get('tflops_per_pass'))"
done | tee gpurun_out/r04l_ab.txt
