#!/bin/bash
cd $GRAFT_REPO_ROOT
TAG=${1:-r03g}
python -c "import torch" >/dev/null 2>&1
timeout 1200 python -m pytest -x -q -m gpu tests/test_hip_kernels.py tests/test_nmn_gpu.py tests/test_nmn_per_module_gpu.py tests/test_modules_gpu.py tests/test_joint_gpu.py tests/test_full_size_gpu.py > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
tail -4 gpurun_out/${TAG}_pytest.log
for V in "PNMN_MASK_BWD_MODE=2" "PNMN_MASK_BWD_MODE=1"; do
  env $V timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('$V', d['value'], d['ms_per_step'], 'conv_nhwc', r['achieved'], r['tflops_per_pass'], {k: (v['tflops'], v['ms_per_step']) for k, v in r['kernels']['conv_nhwc']['by_call_site'].items() if 'module' in k})" | tee -a gpurun_out/${TAG}_ab.txt
  env $V timeout 300 python bench.py --batch 128 --steps 80 --warmup 10 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$V b128', d['value'], d['ms_per_step'], 'host busy', d['host_busy_ms_per_step'], 'blocked', d['host_blocked_ms_per_step'])" | tee -a gpurun_out/${TAG}_ab.txt
done
