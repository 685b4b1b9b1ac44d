cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; python -c "import torch" >/dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tc -o tc -- python scripts/conv_launch_table.py 256 > gpurun_out/r04_tc_table.txt 2>&1
python profiles/summarize.py $(find /tmp/prof_tc -name '*_results.db' | head -1) 2>&1 | grep "conv_stream\|conv_wgrad" | cut -c1-120
grep -c "^conv_nhwc" gpurun_out/r04_tc_table.txt; grep -c "^conv_wgrad" gpurun_out/r04_tc_table.txt
