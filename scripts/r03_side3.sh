#!/bin/bash
cd $GRAFT_REPO_ROOT
for B in 128 256 512 768; do
bash scripts/r03_ab_env.sh r04l $B "PNMN_TRUNK_BEFORE_PRIOR=1" "PNMN_TRUNK_BEFORE_PRIOR=0"
done
