#!/bin/bash
# full GPU test suite + a bench line.   usage: bash scripts/r03_full.sh TAG [bench args]
cd $GRAFT_REPO_ROOT
TAG=${1:-r03}; shift
python -c "import torch" >/dev/null 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
tail -5 gpurun_out/${TAG}_pytest.log
timeout 900 python bench.py "$@" > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -2 gpurun_out/${TAG}_bench.err; cut -c1-300 gpurun_out/${TAG}_bench.json
