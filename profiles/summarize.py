#!/usr/bin/env python3
"""rocprofv3 (ROCm 7.2 writes a rocpd sqlite database) -> the per-kernel table `--stats` would show.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o NAME -- python bench.py ...
    python profiles/summarize.py gpurun_out/prof/NAME_results.db > profiles/rNN_kernel_stats.txt
"""
import glob
import hashlib
import os
import sqlite3
import subprocess
import sys


def source_sha(root=None):
    """16 hex digits over everything a kernel's HBM traffic depends on: csrc/*.hip, csrc/*.h, include/*.h and the
    launch planner (probnmn/runtime/*.py).  bench.py quotes a PMC summary only when this matches its own."""
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "probnmn-clevr_amd")
    files = sorted(glob.glob(os.path.join(pkg, "csrc", "*.hip")) + glob.glob(os.path.join(pkg, "csrc", "*.h"))
                   + glob.glob(os.path.join(root, "include", "*.h")) + glob.glob(os.path.join(pkg, "probnmn", "runtime", "*.py")))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.relpath(f, root).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def provenance():
    """Header lines tying a summary to the sources it was measured on (and to the commit, where git is at hand)."""
    print("# source_sha: %s" % source_sha())
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        head = subprocess.run(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=10)
        if head.returncode == 0:
            print("# commit: %s" % head.stdout.strip())
    except Exception:
        pass


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tab = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0]
    suffix = tab.replace("rocpd_kernel_dispatch", "")
    q = f"""select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start),
                   max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size)
            from rocpd_kernel_dispatch{suffix} d join rocpd_info_kernel_symbol{suffix} s on d.kernel_id = s.id
            group by s.kernel_name order by 3 desc"""
    rows = list(cur.execute(q))
    total = sum(r[2] for r in rows)
    print("# kernel-trace summary of %s" % path)
    provenance()
    print("# total kernel time %.3f ms over %d dispatches" % (total / 1e6, sum(r[1] for r in rows)))
    print("%-72s %8s %12s %11s %10s %10s %6s %5s %5s %5s %7s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct", "vgpr", "agpr", "sgpr", "lds"))
    for r in rows:
        name = r[0].replace(".kd", "")
        print("%-72s %8d %12.3f %11.1f %10.1f %10.1f %6.2f %5d %5d %5d %7d" % (
            name[:72], r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total, r[6] or 0, r[7] or 0, r[8] or 0, r[9] or 0))


def pmc(path, window_json=None):
    """Per-kernel sums of the PMC counters of a `rocprofv3 --kernel-trace --pmc X` run.  ``window_json``: the bench line this
    very run printed (its `roofline` object counts the conv launches of the instrumented passes, the LAST conv launches of
    the process, and their algorithmic bytes): the counters of exactly those launches are summed too, so that traffic and
    algorithmic bytes are quoted on ONE population of launches (`# window` header lines, read by bench.py)."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    tab = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0]
    suffix = tab.replace("rocpd_kernel_dispatch", "")
    q = f"""select s.kernel_name, i.name, count(*), sum(e.value), avg(e.value)
            from rocpd_pmc_event{suffix} e
            join rocpd_kernel_dispatch{suffix} d on d.event_id = e.event_id
            join rocpd_info_kernel_symbol{suffix} s on d.kernel_id = s.id
            join rocpd_info_pmc{suffix} i on i.id = e.pmc_id
            group by s.kernel_name, i.name order by 4 desc"""
    provenance()
    print("# PMC summary of %s (FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts wide" % path)
    print("# coalesced reads at half their bytes -- MI355X_MICROARCH.md, HBM section)")
    print("%-72s %-12s %8s %14s %12s" % ("kernel", "counter", "calls", "sum", "avg/launch"))
    if window_json:
        import json

        line = [l for l in open(window_json) if l.startswith("{")][-1]
        roof = json.loads(line).get("roofline") or {}
        n_win = int(sum(roof.get("launches_per_pass") or []))
        fam = {"conv_nhwc": "conv_stream_kernel"}.get(roof.get("kernel"), roof.get("kernel") or "")
        if n_win and fam:
            rows = list(cur.execute(f"""select i.name, d.start, sum(e.value)
                    from rocpd_pmc_event{suffix} e
                    join rocpd_kernel_dispatch{suffix} d on d.event_id = e.event_id
                    join rocpd_info_kernel_symbol{suffix} s on d.kernel_id = s.id
                    join rocpd_info_pmc{suffix} i on i.id = e.pmc_id
                    where s.kernel_name like '%{fam}%' group by i.name, d.id order by d.start"""))
            for ctr in sorted({r[0] for r in rows}):
                vals = [r[2] for r in rows if r[0] == ctr][-n_win:]
                print("# window %s: last %d launches of %s (the instrumented passes of this run): %s %.2f per launch; "
                      "algorithmic bytes per launch %d" % (roof.get("kernel"), len(vals), fam, ctr, sum(vals) / max(len(vals), 1),
                                                           int(roof.get("algorithmic_bytes_per_launch") or 0)))
    for name, ctr, n, tot, avg in cur.execute(q):
        print("%-72s %-12s %8d %14.1f %12.2f" % (name.replace(".kd", "")[:72], ctr, n, tot, avg))


def mfma(path):
    """MFMA utilisation per kernel from a `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass:
    busy cycles summed over the chip's SIMDs / (GPU-active cycles x 256 CUs x 4 SIMDs).  (ROCm 7.2 ships no gfx950
    derived metrics; this is the gfx94x MfmaUtil formula.  A v_mfma_f32_16x16x4_f32 occupies its SIMD's matrix
    pipe for 32 cycles, so the figure can be checked against the algorithmic MFMA count.)"""
    db = sqlite3.connect(path)
    cur = db.cursor()
    tab = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0]
    suffix = tab.replace("rocpd_kernel_dispatch", "")
    q = f"""select s.kernel_name, i.name, count(*), sum(e.value), count(distinct d.id)
            from rocpd_pmc_event{suffix} e
            join rocpd_kernel_dispatch{suffix} d on d.event_id = e.event_id
            join rocpd_info_kernel_symbol{suffix} s on d.kernel_id = s.id
            join rocpd_info_pmc{suffix} i on i.id = e.pmc_id
            group by s.kernel_name, i.name"""
    agg = {}
    for name, ctr, n, tot, disp in cur.execute(q):
        a = agg.setdefault(name.replace(".kd", ""), {})
        a[ctr] = tot
        a[ctr + "#"] = n / max(disp, 1)  # instances the counter is reported in per dispatch (XCDs / shader engines)
        a["calls"] = disp
    print("# MFMA utilisation of %s" % path)
    provenance()
    print("# util = SQ_VALU_MFMA_BUSY_CYCLES (summed over the chip) / (GPU-active cycles x 1024 SIMDs); GRBM_GUI_ACTIVE")
    print("# is reported once per XCD, so active cycles = its sum / its instances per dispatch")
    print("%-88s %8s %16s %14s %8s" % ("kernel", "calls", "MFMA_BUSY_CYCLES", "active cycles", "util"))
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
        busy, act = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), a.get("GRBM_GUI_ACTIVE", 0.0) / max(a.get("GRBM_GUI_ACTIVE#", 1.0), 1.0)
        if busy <= 0:
            continue
        print("%-88s %8d %16.0f %14.0f %7.1f%%" % (name[:88], a["calls"], busy, act, 100.0 * busy / (act * 1024) if act else 0.0))


def _step_marks(rows):
    """End times of the training iterations: an iteration ends with its clamp_adam_kernel dispatch -- the LARGE one where a
    trainer splits the update over two streams (round 6: the NMN's share on the trunk's stream, a small launch for the
    seq2seq models on the other)."""
    ends = sorted(r[2] for r in rows if "clamp_adam_kernel" in r[0])
    marks = []
    for e in ends:  # launches that end within a millisecond of each other belong to one iteration: its last one marks it
        if marks and e - marks[-1] < 1e6:
            marks[-1] = e
        else:
            marks.append(e)
    return marks


def steady(path, nsteps):
    """Per-STEP kernel table of the last `nsteps` training iterations of the trace (an iteration ends with its
    clamp_adam_kernel dispatch): what the timed region of bench.py looks like without the set-up work."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    tab = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0]
    suffix = tab.replace("rocpd_kernel_dispatch", "")
    rows = list(cur.execute(f"""select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch{suffix} d
                                join rocpd_info_kernel_symbol{suffix} s on d.kernel_id = s.id order by d.start"""))
    marks = _step_marks(rows)
    t0, t1 = marks[-nsteps - 1], marks[-1]
    win = [r for r in rows if r[1] >= t0 and r[2] <= t1]
    agg = {}
    for name, a, b in win:
        e = agg.setdefault(name.replace(".kd", ""), [0, 0])
        e[0] += 1
        e[1] += b - a
    busy, cur_end = 0, t0
    for _, a, b in win:  # union of the kernel intervals (two streams overlap)
        if b > cur_end:
            busy += b - max(a, cur_end)
            cur_end = b
    total = sum(v[1] for v in agg.values())
    print("# last %d iterations of %s: %.3f ms per iteration wall, %.3f ms GPU busy (union), %.3f ms sum of kernels, %d dispatches per iteration"
          % (nsteps, path, (t1 - t0) / 1e6 / nsteps, busy / 1e6 / nsteps, total / 1e6 / nsteps, len(win) // nsteps))
    print("%-88s %10s %12s %10s %6s" % ("kernel", "calls/it", "ms/it", "avg_us", "pct"))
    for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-88s %10.1f %12.4f %10.1f %6.2f" % (name[:88], n / nsteps, t / 1e6 / nsteps, t / 1e3 / n, 100.0 * t / total))


def timeline(path):
    """Every dispatch of the LAST training iteration in start order: offset, duration, queue, kernel."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    tab = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0]
    suffix = tab.replace("rocpd_kernel_dispatch", "")
    cols = [r[1] for r in cur.execute(f"pragma table_info({tab})")]
    qcol = "d.stream_id" if "stream_id" in cols else ("d.queue_id" if "queue_id" in cols else "0")
    rows = list(cur.execute(f"""select s.kernel_name, d.start, d.end, {qcol} from rocpd_kernel_dispatch{suffix} d
                                join rocpd_info_kernel_symbol{suffix} s on d.kernel_id = s.id order by d.start"""))
    marks = _step_marks(rows)
    t0, t1 = marks[-2], marks[-1]
    print("# columns of %s: %s" % (tab, ", ".join(cols)))
    print("# last iteration: %.3f ms" % ((t1 - t0) / 1e6))
    prev_end = {}
    for name, a, b, q in rows:
        if a < t0 or b > t1:
            continue
        gap = (a - prev_end[q]) / 1e3 if q in prev_end else 0.0
        prev_end[q] = b
        short = name.replace(".kd", "").replace("_ZN12_GLOBAL__N_1", "").replace("_ZN2at6native", "at::")[:70]
        print("%9.1f us  +%8.1f us  gap %7.1f  q%-3s %s" % ((a - t0) / 1e3, (b - a) / 1e3, gap, q, short))


if __name__ == "__main__":
    if sys.argv[1] == "--mfma":
        mfma(sys.argv[2])
    elif sys.argv[1] == "--timeline":
        timeline(sys.argv[2])
    elif sys.argv[1] == "--pmc":
        pmc(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
    elif sys.argv[1] == "--steady":
        steady(sys.argv[3], int(sys.argv[2]))
    else:
        main(sys.argv[1])
