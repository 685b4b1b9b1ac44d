#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest -x -q -m gpu tests/test_nmn_gpu.py tests/test_joint_gpu.py tests/test_full_size_gpu.py 2>&1 | tail -3
bash scripts/r03_ab_env.sh r04o 1024 "PNMN_SPLIT_FC_BACKWARD=0" "PNMN_SPLIT_FC_BACKWARD=1"
bash scripts/r03_ab_env.sh r04o 128 "PNMN_SPLIT_FC_BACKWARD=0" "PNMN_SPLIT_FC_BACKWARD=1"
