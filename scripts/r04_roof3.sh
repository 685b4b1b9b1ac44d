#!/bin/bash
# run-to-run variation of bench.py's roofline object (the sampled programs of the instrumented steps differ)
cd $GRAFT_REPO_ROOT
python -c "import torch" >/dev/null 2>&1
for i in 1 2 3; do
timeout 500 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
r=d['roofline']
print(d['ms_per_step'], r['frac'], r['launches_per_step'], r['tflops_per_pass'], d['config']['module_primitives_per_step'])"
done
