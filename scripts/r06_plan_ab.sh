# A/B of the plan's launch widths at 128 questions (encoder passes per wavefront launch forward / backward, decoder passes per
# backward launch, workgroups per GEMM launch).  usage: bash scripts/r06_plan_ab.sh TAG "cfg;cfg;..."  (cfg = "pg fwd bwd dec")
cd $GRAFT_REPO_ROOT
TAG=${1:-r06ab}
CFGS=${2:-"0 0 0 3;1 0 0 3;1 1 0 3;1 2 0 3;1 0 1 3;1 2 1 3;0 2 0 3"}
OUT=gpurun_out/${TAG}_plan_ab.txt
: > $OUT
IFS=';'
for rep in 1 2; do
for cfg in $CFGS; do
  IFS=' ' read -r a b c d <<< "$cfg"
  PNMN_PLAN_PG_ENC=$a PNMN_PLAN_FWD_ENC=$b PNMN_PLAN_BWD_ENC=$c PNMN_PLAN_DEC_GROUP=$d python bench.py --batch 128 --steps 80 --warmup 10 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pg_enc $a fwd_enc $b bwd_enc $c dec_group $d: %.3f ms  host busy %.2f' % (d['ms_per_step'], d['host_busy_ms_per_step']))" >> $OUT
done
done
sort $OUT
