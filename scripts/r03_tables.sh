#!/bin/bash
# per-launch table of the headline step's conv calls + the b128 steady profile.  usage: bash scripts/r03_tables.sh TAG
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import torch" >/dev/null 2>&1
cd $R
TAG=${1:-r03c}
PNMN_LAUNCH_TABLE=gpurun_out/${TAG}_launch_table.txt timeout 600 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-extras --roofline-passes 1 > gpurun_out/${TAG}_lt_bench.json 2> gpurun_out/${TAG}_lt_bench.err
bash scripts/steady_profile.sh ${TAG}_b128 --batch 128 --steps 40 --warmup 5
