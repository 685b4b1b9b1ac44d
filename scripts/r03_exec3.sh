#!/bin/bash
# where the executor's time goes: per-iteration kernel tables of the 1024-question step with and without it
cd $GRAFT_REPO_ROOT
for E in 0 1; do
  PNMN_TRUNK_EXEC=$E bash scripts/steady_profile.sh r04h_exec$E --steps 12 --warmup 4 >/dev/null 2>&1
  echo "== PNMN_TRUNK_EXEC=$E"; head -30 gpurun_out/r04h_exec${E}_steady.txt
done
