#!/bin/bash
# HBM read traffic (FETCH_SIZE) of isolated module-conv launches under different item -> XCD mappings.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import torch" >/dev/null 2>&1
cd $R
i=0
for cfg in "NW=15" "NW=15 SORTED=1" "NW=15 SORTED=1 PNMN_CONV_XCD_RANGES=1" "NW=1"; do
  i=$((i+1))
  env $cfg timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmcc_$i -o p -- python scripts/conv_modes.py 1 > /dev/null 2>&1
  echo "== $cfg"
  python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob('/tmp/pmcc_$i/**/*_results.db', recursive=True)[0]); cur = db.cursor()
tab = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0]
sfx = tab.replace('rocpd_kernel_dispatch', '')
q = f"""select d.grid_size_x, count(*), avg(e.value) from rocpd_pmc_event{sfx} e join rocpd_kernel_dispatch{sfx} d on d.event_id = e.event_id
        join rocpd_info_kernel_symbol{sfx} s on d.kernel_id = s.id where s.kernel_name like '%conv_nhwc%' group by d.grid_size_x order by 1"""
for g, n, v in cur.execute(q):
    items = g // 512
    print("  items %5d launches %3d  FETCH %.1f MB/launch (x2 corrected)  = %.0f KB/item   (in+out alg 200 KB/item, weights 590 KB each)" % (items, n, v * 2 * 1024 / 1e6, v * 2 * 1024 / 1e3 / max(items, 1)))
PY
done
