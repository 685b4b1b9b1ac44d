# 128-question step: alone / behind the other sides, host profile, steady kernel table.   usage: bash scripts/r05_b128_diag.sh TAG
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import torch" >/dev/null 2>&1
cd $R
TAG=${1:-r05b}
for i in 1 2; do
  python bench.py --batch 128 --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('alone headline b128', d['ms_per_step'], d.get('host_busy_ms_per_step'), d.get('host_blocked_ms_per_step'))"
done > gpurun_out/${TAG}_b128_alone.txt 2>&1
python bench.py --no-cpu-baseline --no-roofline --sides joint_training_b128 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('side only', d['joint_training_b128'])" >> gpurun_out/${TAG}_b128_alone.txt 2>&1
timeout 300 python scripts/host_cprofile.py 128 > gpurun_out/${TAG}_b128_cprofile.txt 2>&1
timeout 300 python scripts/r04_host_ops.py 128 > gpurun_out/${TAG}_b128_host_ops.txt 2>&1
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_$TAG -o $TAG -- python bench.py --batch 128 --steps 40 --warmup 10 --no-cpu-baseline --no-extras --no-roofline > gpurun_out/${TAG}_b128_prof.log 2>&1
python profiles/summarize.py --steady 30 $(find /tmp/prof_$TAG -name '*_results.db' | head -1) > gpurun_out/${TAG}_b128_steady.txt 2>&1
cat gpurun_out/${TAG}_b128_alone.txt
