# steady kernel table of the question_coding side object (the last iterations of the process).   usage: bash scripts/r05_qc_steady.sh TAG
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import torch" >/dev/null 2>&1
cd $R
TAG=${1:-r05qc}
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_$TAG -o $TAG -- python bench.py --no-cpu-baseline --no-roofline --sides question_coding > gpurun_out/${TAG}_prof.log 2>&1
python profiles/summarize.py --steady 8 $(find /tmp/prof_$TAG -name '*_results.db' | head -1) > gpurun_out/${TAG}_qc_steady.txt 2>&1
head -45 gpurun_out/${TAG}_qc_steady.txt | cut -c1-70,88-140
