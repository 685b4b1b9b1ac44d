#!/bin/bash
# trunk executor: the GPU suites that run the trunk, with the executor on; every step under its own timeout
cd $GRAFT_REPO_ROOT
python -c "import torch" >/dev/null 2>&1
export PNMN_TRUNK_EXEC=1 PNMN_EXEC_DEBUG=1
for f in tests/test_nmn_gpu.py tests/test_joint_gpu.py tests/test_modules_gpu.py tests/test_full_size_gpu.py tests/test_dp_trainers_gpu.py tests/test_trainers_gpu.py; do
  [ -f $f ] || continue
  echo "== $f"; timeout 400 python -m pytest -x -q -m gpu $f 2>&1 | tail -4
done
