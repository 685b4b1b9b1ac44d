#!/bin/bash
# round-3 GPU check: the tests touched this round first, then a bench line.   usage: bash scripts/r03_check.sh <tag> [pytest args]
cd $GRAFT_REPO_ROOT
TAG=${1:-r03a}; shift
python -c "import torch" >/dev/null 2>&1
timeout 1500 python -m pytest -x -q -m gpu "$@" > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
tail -15 gpurun_out/${TAG}_pytest.log
