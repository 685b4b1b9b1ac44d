#!/bin/bash
# round record: full GPU suite, the driver's bench command, the rocprofv3 round profile, b128 steady table + step timeline
cd $GRAFT_REPO_ROOT
TAG=${1:-r03a}
python -c "import torch" >/dev/null 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
tail -4 gpurun_out/${TAG}_pytest.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/${TAG}_smoke.log 2>&1; tail -2 gpurun_out/${TAG}_smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -2 gpurun_out/${TAG}_bench.err; cut -c1-200 gpurun_out/${TAG}_bench.json
bash scripts/profile_round.sh $TAG
bash scripts/steady_profile.sh ${TAG}_b128 --batch 128 --steps 40 --warmup 5
timeout 300 python scripts/step_timeline.py 128 40 --free > gpurun_out/${TAG}_b128_step_timeline_free.txt 2>&1
timeout 300 python scripts/recurrent_rate.py > gpurun_out/${TAG}_recurrent_rate.txt 2>&1
# the trunk executor (opt-in): dispatches per iteration and step time of the 128-question step with it
PNMN_TRUNK_EXEC=1 bash scripts/steady_profile.sh ${TAG}_b128_exec --batch 128 --steps 40 --warmup 5
