#!/bin/bash
# A/B of two checkouts of the python package in ONE gpurun call (box-to-box variance of the host-bound side
# objects is larger than most changes): scripts/_ab/old (the baseline commit, git-ignored; make it here with
#   mkdir -p scripts/_ab/old && git archive <commit> probnmn-clevr_amd/probnmn | tar -x -C scripts/_ab/old
# -- it runs against the tree's library, so the baseline must not need symbols the tree dropped) vs the tree.
#   usage: bash scripts/ab.sh [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -c "import torch" >/dev/null 2>&1
for i in $(seq 1 ${1:-2}); do
  for which in old new; do
    if [ $which = old ]; then export PNMN_PKG_DIR=$R/scripts/_ab/old/probnmn-clevr_amd PNMN_LIB=$R/probnmn-clevr_amd/lib/libprobnmn_hip.so; else unset PNMN_PKG_DIR PNMN_LIB; fi
    timeout 600 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('$which', 'headline %.2f ms' % d['ms_per_step'], ' '.join('%s %.2f ms (host %.2f)' % (k[:12], d[k]['ms_per_step'], d[k]['host_busy_ms_per_step']) for k in ('joint_training_b128','question_coding','module_training','joint_training_28x28')))"
  done
done
