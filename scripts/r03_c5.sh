#!/bin/bash
cd $GRAFT_REPO_ROOT
python -c "import torch" >/dev/null 2>&1
for rep in 1 2; do
for V in "PNMN_SHARED_CONV_CUS=192" "PNMN_SHARED_CONV_CUS=224" "PNMN_SHARED_CONV_CUS=240" "PNMN_SHARED_CONV_CUS=256"; do
  env $V TAGV="$V" timeout 300 python scripts/r03_c5.py 2>&1 | grep "^c5" | tee -a gpurun_out/r04r_c5.txt
done
done
