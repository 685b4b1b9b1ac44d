#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of every conv_stream dispatch of one module_training step at 1024 questions, largest first:
# which launches carry the HBM traffic beyond the algorithmic bytes?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; python -c "import torch" >/dev/null 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmcd_$C -o p -- python scripts/conv_launch_table.py 1024 > gpurun_out/r04_pmcd_$C.log 2>&1
  python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob('/tmp/pmcd_$C/**/*_results.db', recursive=True)[0]); cur = db.cursor()
tab = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0]
sfx = tab.replace('rocpd_kernel_dispatch', '')
q = f"""select s.kernel_name, d.start, d.end, e.value from rocpd_pmc_event{sfx} e join rocpd_kernel_dispatch{sfx} d on d.event_id = e.event_id
        join rocpd_info_kernel_symbol{sfx} s on d.kernel_id = s.id where s.kernel_name like '%conv_stream%' order by d.start"""
rows = list(cur.execute(q))
# aggregate the counter's per-XCD / per-instance rows of one dispatch
agg = {}
for name, a, b, v in rows:
    k = (name, a, b)
    agg[k] = agg.get(k, 0.0) + v
disp = sorted(agg.items(), key=lambda kv: kv[0][1])
n = len(disp) // 4  # four steps in the script: take the last one
last = disp[-n:]
f = 2.0 if "$C" == "FETCH_SIZE" else 1.0
print("$C: %d dispatches in the last step, total %.1f MB" % (n, sum(v for _, v in last) * f * 1024 / 1e6))
for (name, a, b), v in sorted(last, key=lambda kv: -kv[1])[:12]:
    print("   %8.1f us  %9.1f MB  %s" % ((b - a) / 1e3, v * f * 1024 / 1e6, "1x1" if "ELi1EEE" in name else "3x3"))
print("  in issue order (dispatch index: MB):", " ".join("%d%s:%.0f" % (i, "*" if "ELi1EEE" in k[0] else "", v * f * 1024 / 1e6) for i, (k, v) in enumerate(last)))
PY
done
grep "^conv_nhwc" gpurun_out/r04_pmcd_FETCH_SIZE.log | awk '{print NR-1": "$0}' | cut -c1-90
