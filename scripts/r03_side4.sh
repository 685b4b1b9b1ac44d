#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scripts/r03_ab_env.sh r04m 1024 "PNMN_X=0" "PNMN_NMN_STREAM_PRIORITY=-1"
bash scripts/r03_ab_env.sh r04m 128 "PNMN_X=0" "PNMN_NMN_STREAM_PRIORITY=-1"
timeout 300 python scripts/step_timeline.py 1024 20 --free > gpurun_out/r04m_b1024_step_timeline_free.txt 2>&1
tail -40 gpurun_out/r04m_b1024_step_timeline_free.txt
