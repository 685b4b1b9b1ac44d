# A/B of environment switches on the 128-question joint step (and others).  usage: bash scripts/r06_env_ab.sh TAG "ENV=a ENV2=b;ENV=c" [bench args]
cd $GRAFT_REPO_ROOT
TAG=${1:-r06ab}; CFGS=$2; shift; shift
ARGS="$*"
[ -z "$ARGS" ] && ARGS="--batch 128 --steps 80 --warmup 10"
OUT=gpurun_out/${TAG}_env_ab.txt
: > $OUT
IFS=';' read -ra LIST <<< "$CFGS"
for rep in 1 2 3; do
for cfg in "${LIST[@]}"; do
  env $cfg python bench.py $ARGS --no-cpu-baseline --no-roofline --no-extras 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %.3f ms  host busy %.2f' % ('$cfg', d['ms_per_step'], d['host_busy_ms_per_step']))" >> $OUT
done
done
sort $OUT
