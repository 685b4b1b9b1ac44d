#!/bin/bash
# LDS bank conflicts of the streamed conv kernel (isolated launches)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import torch" >/dev/null 2>&1
cd $R
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pmcl -o p -- python scripts/r04_cycles.py 1024 0 > /dev/null 2>&1
python profiles/summarize.py --pmc $(find /tmp/pmcl -name '*_results.db' | head -1) 2>&1 | grep -i "conv_" | sed "s/^[^ ]* //"
