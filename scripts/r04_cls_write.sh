cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; python -c "import torch" >/dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmcw -o p -- python scripts/r04_cls_write.py 2>&1 | grep "algorithmic"
python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob('/tmp/pmcw/**/*_results.db', recursive=True)[0]); cur = db.cursor()
tab = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0]
sfx = tab.replace('rocpd_kernel_dispatch', '')
q = f"""select d.start, e.value from rocpd_pmc_event{sfx} e join rocpd_kernel_dispatch{sfx} d on d.event_id = e.event_id
        join rocpd_info_kernel_symbol{sfx} s on d.kernel_id = s.id where s.kernel_name like '%conv_stream%' order by d.start"""
agg = {}
for a, v in cur.execute(q):
    agg[a] = agg.get(a, 0.0) + v
print("WRITE_SIZE per conv dispatch (MB):", " ".join("%.1f" % (v * 1024 / 1e6) for _, v in sorted(agg.items())))
PY
