cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import torch" >/dev/null 2>&1
cd $R
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_g -o g -- python bench.py --steps 30 --warmup 2 --no-cpu-baseline --no-extras --no-roofline > gpurun_out/g.log 2>&1
python profiles/gaps.py $(find /tmp/prof_g -name '*_results.db' | head -1) 0.5 > gpurun_out/gaps.txt 2>&1
