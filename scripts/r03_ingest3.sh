#!/bin/bash
cd $GRAFT_REPO_ROOT
TAG=${1:-r03k}
python -c "import torch" >/dev/null 2>&1
for G in 4 8 12; do PNMN_INGEST_WGS=$G timeout 300 python scripts/ingest_rate.py 14 2>&1 | grep 'gather 1024'; done
for M in "PNMN_INGEST=kernel PNMN_INGEST_WGS=4" "PNMN_INGEST=kernel PNMN_INGEST_WGS=8" "PNMN_INGEST=kernel PNMN_INGEST_WGS=12"; do
env $M timeout 600 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-roofline --no-extras-but-ingest 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
i = d['joint_training_ingest']
print('$M resident', d['value'], d['ms_per_step'], 'ingest', {k: i.get(k) for k in ('value','ms_per_step','pcie_GBs_per_gpu','slowdown_vs_resident','error')})" | tee -a gpurun_out/${TAG}_ingest_step.txt
done
