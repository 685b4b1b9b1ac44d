#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scripts/r03_ab_env.sh r04u 512 "PNMN_STEM_AFTER_ENCODE=0" "PNMN_STEM_AFTER_ENCODE=2"
bash scripts/r03_ab_env.sh r04u 256 "PNMN_STEM_AFTER_ENCODE=0" "PNMN_STEM_AFTER_ENCODE=2"
bash scripts/r03_ab_env.sh r04u 1024 "PNMN_X=1"
timeout 300 python -m pytest -x -q -m gpu tests/test_joint_gpu.py tests/test_full_size_gpu.py tests/test_dp_trainers_gpu.py 2>&1 | tail -2
