#!/usr/bin/env python3
"""Idle gaps on the GPU timeline of a rocprofv3 kernel trace (rocpd sqlite): the largest intervals
between the end of one dispatch and the start of the next, with the kernels on either side.

    python profiles/gaps.py /tmp/prof/x_results.db [first_fraction]   (analyses the last 40% of the trace)
"""
import sqlite3
import sys


def main(path, tail=0.4):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tab = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0]
    suffix = tab.replace("rocpd_kernel_dispatch", "")
    rows = list(cur.execute(f"""select d.start, d.end, s.kernel_name from rocpd_kernel_dispatch{suffix} d
                                join rocpd_info_kernel_symbol{suffix} s on d.kernel_id = s.id order by d.start"""))
    rows = rows[int(len(rows) * (1 - tail)):]
    span = rows[-1][1] - rows[0][0]
    busy_until, gaps, busy = rows[0][1], [], 0
    for i in range(1, len(rows)):
        st, en, name = rows[i]
        if st > busy_until:
            gaps.append((st - busy_until, rows[i - 1][2], name))
        busy_until = max(busy_until, en)
    idle = sum(g[0] for g in gaps)
    print("# %d dispatches over %.2f ms; idle %.2f ms (%.1f%%) in %d gaps" % (len(rows), span / 1e6, idle / 1e6, 100.0 * idle / span, len(gaps)))
    agg = {}
    for g, a, b in gaps:
        k = (a.replace(".kd", "")[:48], b.replace(".kd", "")[:48])
        t = agg.setdefault(k, [0, 0])
        t[0] += g
        t[1] += 1
    # the individual gaps of the last ~60 ms, in time order (one step of the joint bench)
    t_end = rows[-1][1]
    print("# gaps > 100 us in the last 60 ms (time before the end of the trace, ms):")
    busy_until = rows[0][1]
    for i in range(1, len(rows)):
        st, en, name = rows[i]
        if st > busy_until and st - busy_until > 100e3 and t_end - st < 60e6:
            print("  t-%.2f  idle %.3f ms   after %-40s before %-40s" % ((t_end - st) / 1e6, (st - busy_until) / 1e6,
                                                                        rows[i - 1][2][:40], name[:40]))
        busy_until = max(busy_until, en)
    print("%-50s %-50s %6s %10s %9s" % ("after", "before", "count", "total_ms", "avg_us"))
    for (a, b), (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:30]:
        print("%-50s %-50s %6d %10.3f %9.1f" % (a, b, n, t / 1e6, t / n / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.4)
