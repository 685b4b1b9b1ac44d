#!/bin/bash
# round 4: streamed conv kernel -- kernel parity tests, then isolated launch rates old vs new
cd $GRAFT_REPO_ROOT
python -c "import torch" >/dev/null 2>&1
export PNMN_CONV_STREAM=1
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -k "conv" 2>&1 | tail -15 > gpurun_out/r04a_tests.txt
cat gpurun_out/r04a_tests.txt
for m in auto 1 2 4; do
  PNMN_CONV_STREAM=0 timeout 300 python scripts/conv_modes.py $m 2>&1 | tail -1 | sed 's/^/old /'
  PNMN_CONV_STREAM=1 timeout 300 python scripts/conv_modes.py $m 2>&1 | tail -1 | sed 's/^/new /'
done | tee gpurun_out/r04a_modes.txt
