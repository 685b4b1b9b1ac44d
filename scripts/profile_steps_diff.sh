cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import torch" >/dev/null 2>&1
cd $R
for N in 2 22; do
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_s$N -o s$N -- python bench.py --steps $N --warmup 1 --no-cpu-baseline --no-extras --no-roofline > gpurun_out/s$N.log 2>&1
python profiles/summarize.py $(find /tmp/prof_s$N -name '*_results.db' | head -1) > gpurun_out/steps${N}_kernel_stats.txt 2>&1
done
