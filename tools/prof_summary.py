"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / avg / total, GPU-busy vs span.

    rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- <cmd>
    python tools/prof_summary.py /tmp/prof/run_results.db [top_n] > profiles/<name>.txt
"""
import glob
import sqlite3
import sys


def main():
    path = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    if not path.endswith(".db"):
        cands = glob.glob(path + "/**/*.db", recursive=True)
        path = cands[0]
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
                       "from kernels group by name order by 6 desc").fetchall()
    tot = sum(r[5] for r in rows)
    n = sum(r[1] for r in rows)
    t0, t1 = cur.execute("select min(start), max(end) from kernels").fetchone()
    # union of the kernel intervals = time with at least one kernel running (streams overlap, so <= the sum)
    active, cur_s, cur_e = 0, None, None
    for s_, e_ in cur.execute("select start, end from kernels order by start"):
        if cur_e is None or s_ > cur_e:
            if cur_e is not None:
                active += cur_e - cur_s
            cur_s, cur_e = s_, e_
        elif e_ > cur_e:
            cur_e = e_
    if cur_e is not None:
        active += cur_e - cur_s
    print(f"# {path}")
    print(f"# kernel launches: {n}   sum of kernel durations: {tot / 1e6:.2f} ms   time with a kernel running (union): "
          f"{active / 1e6:.2f} ms   first-to-last span: {(t1 - t0) / 1e6:.2f} ms")
    print(f"# {'kernel':<96} {'calls':>7} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'total_ms':>10} {'%':>6}")
    for name, cnt, avg, mn, mx, sm in rows[:top]:
        print(f"{name[:98]:<98} {cnt:>7d} {avg / 1e3:>10.2f} {mn / 1e3:>10.2f} {mx / 1e3:>10.2f} {sm / 1e6:>10.2f} "
              f"{100 * sm / tot:>6.2f}")
    if len(rows) > top:
        rest = rows[top:]
        print(f"{'(other %d kernels)' % len(rest):<98} {sum(r[1] for r in rest):>7d} {'':>10} {'':>10} {'':>10} "
              f"{sum(r[5] for r in rest) / 1e6:>10.2f} {100 * sum(r[5] for r in rest) / tot:>6.2f}")


if __name__ == "__main__":
    main()
