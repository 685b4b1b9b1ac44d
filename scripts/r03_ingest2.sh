#!/bin/bash
cd $GRAFT_REPO_ROOT
TAG=${1:-r03i}
python -c "import torch" >/dev/null 2>&1
timeout 600 python -m pytest -x -q -m gpu tests/test_feature_store_gpu.py tests/test_seqglue_gpu.py tests/test_seq2seq_gpu.py tests/test_nmn_gpu.py tests/test_joint_gpu.py tests/test_seq2seq_overrides.py 2>&1 | tail -3
for M in dma kernel; do
PNMN_INGEST=$M PNMN_INGEST_WGS=32 timeout 600 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
i = d['joint_training_ingest']
print('$M resident', d['value'], d['ms_per_step'], 'b128', d['joint_training_b128']['ms_per_step'], 'ingest', {k: i.get(k) for k in ('value','ms_per_step','pcie_GBs_per_gpu','slowdown_vs_resident','error')})" | tee -a gpurun_out/${TAG}_ingest_step.txt
done
