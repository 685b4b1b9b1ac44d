# question_coding (512 rows) steady kernel table + timeline.  usage: bash scripts/r06_qc_profile.sh TAG [env...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import torch" >/dev/null 2>&1
cd $R
TAG=${1:-r06qc}; shift
env "$@" timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_$TAG -o $TAG -- python bench.py --no-cpu-baseline --no-roofline --sides question_coding --steps 2 --warmup 1 --settle 2 > gpurun_out/${TAG}_bench.log 2>&1
python profiles/summarize.py --steady 8 $(find /tmp/prof_$TAG -name '*_results.db' | head -1) > gpurun_out/${TAG}_steady.txt 2>&1
python profiles/summarize.py --timeline $(find /tmp/prof_$TAG -name '*_results.db' | head -1) > gpurun_out/${TAG}_timeline.txt 2>&1
head -40 gpurun_out/${TAG}_steady.txt | cut -c1-150
