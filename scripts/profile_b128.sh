# b128 joint step: kernel table, idle gaps on the GPU timeline, host profile.   usage: bash scripts/profile_b128.sh r02a [batch]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import torch" >/dev/null 2>&1
cd $R
TAG=${1:-r02a}
B=${2:-128}
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o $TAG -- python bench.py --batch $B --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-roofline > gpurun_out/${TAG}_b${B}.log 2>&1
DB=$(find /tmp/prof_$TAG -name '*_results.db' | head -1)
python profiles/summarize.py $DB > gpurun_out/${TAG}_b${B}_kernel_stats.txt 2>&1
python profiles/gaps.py $DB 0.5 > gpurun_out/${TAG}_b${B}_gaps.txt 2>&1
python profiles/summarize.py --steady 20 $DB > gpurun_out/${TAG}_b${B}_steady.txt 2>&1
python profiles/summarize.py --timeline $DB > gpurun_out/${TAG}_b${B}_timeline.txt 2>&1
timeout 300 python scripts/joint_host_profile.py $B > gpurun_out/${TAG}_b${B}_host.txt 2>&1
