# the 128-question side alone / behind the other sides / with the earlier sides' objects frozen out of the collector
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import torch" >/dev/null 2>&1
cd $R
TAG=${1:-r05c}
get() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['joint_training_b128']; print('$1', b['ms_per_step'], b['host_busy_ms_per_step'], b['host_blocked_ms_per_step'])"; }
SIDES=joint_training_28x28,joint_training_ingest,evaluate_answer_accuracy,train_validate_train,joint_training_b128
{
python bench.py --no-cpu-baseline --no-roofline --sides joint_training_b128 2>/dev/null | get alone
python bench.py --no-cpu-baseline --no-roofline --sides $SIDES 2>/dev/null | get behind_all
PNMN_BENCH_GC=freeze python bench.py --no-cpu-baseline --no-roofline --sides $SIDES 2>/dev/null | get behind_all_frozen
python bench.py --no-cpu-baseline --no-roofline --sides joint_training_28x28,joint_training_b128 2>/dev/null | get behind_28
python bench.py --no-cpu-baseline --no-roofline --sides joint_training_ingest,joint_training_b128 2>/dev/null | get behind_ingest
python bench.py --no-cpu-baseline --no-roofline --sides evaluate_answer_accuracy,train_validate_train,joint_training_b128 2>/dev/null | get behind_eval
} > gpurun_out/${TAG}_b128_order.txt 2>&1
cat gpurun_out/${TAG}_b128_order.txt
PNMN_LAUNCH_TABLE=gpurun_out/${TAG}_launch_table.txt python bench.py --no-cpu-baseline --no-extras --roofline-passes 2 > gpurun_out/${TAG}_bench_rf.json 2>/dev/null
python -m pytest tests/test_optim.py tests/test_hip_kernels.py tests/test_trajectory_gpu.py tests/test_abi.py -m gpu -x -q 2>&1 | tail -3
