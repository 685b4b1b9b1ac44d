#!/bin/bash
cd $GRAFT_REPO_ROOT
TAG=${1:-r03h}
python -c "import torch" >/dev/null 2>&1
timeout 600 python -m pytest -x -q -m gpu tests/test_feature_store_gpu.py 2>&1 | tail -3
for G in 32 64 128 256; do echo "PNMN_INGEST_WGS=$G"; PNMN_INGEST_WGS=$G timeout 300 python scripts/ingest_rate.py 14 2>&1 | grep gather; done | tee gpurun_out/${TAG}_ingest_rate.txt
PNMN_INGEST_WGS=64 timeout 300 python scripts/ingest_rate.py 28 2>&1 | grep gather | tee -a gpurun_out/${TAG}_ingest_rate.txt
for G in 32 64 128; do
PNMN_INGEST_WGS=$G timeout 600 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('WGS=$G resident', d['value'], d['ms_per_step'], 'ingest', d['joint_training_ingest'])" | tee -a gpurun_out/${TAG}_ingest_step.txt
done
for B in 64 128 512; do timeout 200 python scripts/fc_gemm_probe.py $B 2>&1 | sed "s/^/B=$B /"; done | tee gpurun_out/${TAG}_fc_probe.txt
