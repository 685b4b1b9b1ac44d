#!/bin/bash
# Per-iteration kernel table of the timed steps only.  usage: bash scripts/steady_profile.sh TAG [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import torch" >/dev/null 2>&1
cd $R
TAG=${1:-steady}; shift
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_$TAG -o $TAG -- python bench.py --no-cpu-baseline --no-extras --no-roofline "$@" > gpurun_out/${TAG}_bench.log 2>&1
python profiles/summarize.py --steady 16 $(find /tmp/prof_$TAG -name '*_results.db' | head -1) > gpurun_out/${TAG}_steady.txt 2>&1
python profiles/summarize.py --timeline $(find /tmp/prof_$TAG -name '*_results.db' | head -1) > gpurun_out/${TAG}_timeline.txt 2>&1
head -3 gpurun_out/${TAG}_steady.txt
