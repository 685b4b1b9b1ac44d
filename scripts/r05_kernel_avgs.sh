# average duration of the recurrent kernels in the headline step (single-stream schedule, rocprofv3 kernel trace).   usage: bash scripts/r05_kernel_avgs.sh TAG
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import torch" >/dev/null 2>&1
cd $R
TAG=${1:-r05j}
PNMN_NMN_STREAM=0 timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_$TAG -o $TAG -- python bench.py --no-cpu-baseline --no-extras --no-roofline > gpurun_out/${TAG}_prof.log 2>&1
python profiles/summarize.py --steady 20 $(find /tmp/prof_$TAG -name '*_results.db' | head -1) > gpurun_out/${TAG}_steady_single_stream.txt 2>&1
head -24 gpurun_out/${TAG}_steady_single_stream.txt | cut -c1-70,88-140
