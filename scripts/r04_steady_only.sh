cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; python -c "import torch" >/dev/null 2>&1
TAG=r04p
PNMN_NMN_STREAM=0 timeout 900 rocprofv3 --kernel-trace -d /tmp/prof3_$TAG -o $TAG -- python bench.py --no-cpu-baseline --no-extras --no-roofline > gpurun_out/${TAG}_prof3_bench.log 2>&1
python profiles/summarize.py --steady 20 $(find /tmp/prof3_$TAG -name '*_results.db' | head -1) > gpurun_out/${TAG}_steady_single_stream.txt 2>&1
head -8 gpurun_out/${TAG}_steady_single_stream.txt | cut -c1-160
