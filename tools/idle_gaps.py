"""GPU-busy accounting of the loop window of a rocprofv3 kernel trace taken over tools/loop_trace.py: everything after
the last pause > 0.5 s.  Prints busy (union of kernel intervals) vs span per step, the idle time split by gap length,
the largest gaps with the kernels on either side, and the kernel table of the window.

    python tools/idle_gaps.py <dir-or-db> <steps> [top_n]
"""
import glob
import sqlite3
import sys


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    if not path.endswith(".db"):
        path = glob.glob(path + "/**/*.db", recursive=True)[0]
    cur = sqlite3.connect(path).cursor()
    ks = cur.execute("select start, end, name from kernels order by start").fetchall()
    cut, last_end = 0, ks[0][1]
    for i, (s, e, _) in enumerate(ks):
        if s - last_end > 5e8:
            cut = i
        last_end = max(last_end, e)
    ks = ks[cut:]
    t0, t1 = ks[0][0], max(k[1] for k in ks)
    gaps, busy, cur_s, cur_e, prev_name = [], 0, ks[0][0], ks[0][1], ks[0][2]
    for s, e, name in ks[1:]:
        if s > cur_e:
            gaps.append((s - cur_e, cur_e - t0, prev_name, name))
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
            prev_name = name
        elif e > cur_e:
            cur_e = e
            prev_name = name
    busy += cur_e - cur_s
    span = t1 - t0
    tot = sum(e - s for s, e, _ in ks)
    print(f"# window: {len(ks)} launches over {steps} steps; per step: span {span / steps / 1e6:.2f} ms, busy (union) "
          f"{busy / steps / 1e6:.2f} ms, idle {(span - busy) / steps / 1e6:.2f} ms, sum of kernel durations "
          f"{tot / steps / 1e6:.2f} ms, launches {len(ks) / steps:.0f}")
    for lo, hi in ((0, 5e3), (5e3, 2e4), (2e4, 1e5), (1e5, 1e6), (1e6, 1e12)):
        sel = [g[0] for g in gaps if lo <= g[0] < hi]
        print(f"# idle gaps {lo / 1e3:>6.0f}-{hi / 1e3:<8.0f} us: {len(sel) / steps:>8.1f} per step, "
              f"{sum(sel) / steps / 1e6:>7.2f} ms per step")
    print("# largest gaps: ms, at ms-into-window, kernel before -> kernel after")
    for g in sorted(gaps, reverse=True)[:top]:
        print(f"{g[0] / 1e6:8.3f} {g[1] / 1e6:9.2f}  {g[2][:60]}  ->  {g[3][:60]}")
    agg = {}
    for s, e, name in ks:
        a = agg.setdefault(name, [0, 0])
        a[0] += 1
        a[1] += e - s
    print(f"# {'kernel':<98} {'calls/step':>10} {'avg_us':>10} {'ms/step':>10} {'%':>6}")
    for name, (c, sm) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{name[:98]:<98} {c / steps:>10.1f} {sm / c / 1e3:>10.2f} {sm / steps / 1e6:>10.2f} {100 * sm / tot:>6.2f}")
    by_grid(cur, t0, steps, 60)
    by_queue(cur, t0, steps)
    timeline(cur, t0, steps)


def by_grid(cur, t_from, steps, top):
    """Per (kernel, grid) totals inside the window — which launch shapes carry a kernel's time."""
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    gcol = next((c for c in ("grid_x", "grid_size_x", "grid_size") if c in cols), None)
    if gcol is None:
        print("# (no grid column in the kernels view:", cols, ")")
        return
    rows = cur.execute(f"select name, {gcol}, count(*), sum(end-start) from kernels where start >= ? group by name, {gcol} "
                       "order by 4 desc", (t_from,)).fetchall()
    print(f"# {'kernel / grid (threads)':<98} {'calls/step':>10} {'avg_us':>10} {'ms/step':>10}")
    for name, grid, c, sm in rows[:top]:
        print(f"{(name[:80] + ' / ' + str(grid)):<98} {c / steps:>10.1f} {sm / c / 1e3:>10.2f} {sm / steps / 1e6:>10.2f}")


def by_queue(cur, t_from, steps):
    """Busy time per HIP stream (hardware queue) inside the window: what each stream carries and how much overlaps."""
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    qcol = next((c for c in ("stream_id", "queue_id", "stream", "queue") if c in cols), None)
    if qcol is None:
        print("# (no stream/queue column in the kernels view:", cols, ")")
        return
    print(f"# per {qcol}: launches/step, sum of kernel durations ms/step, top kernels")
    for (q,) in cur.execute(f"select distinct {qcol} from kernels where start >= ?", (t_from,)).fetchall():
        rows = cur.execute(f"select name, count(*), sum(end-start) from kernels where start >= ? and {qcol} = ? group by name "
                           "order by 3 desc", (t_from, q)).fetchall()
        n, tot = sum(r[1] for r in rows), sum(r[2] for r in rows)
        tops = ", ".join(f"{r[0].split('(')[0].split('::')[-1][:28]} {r[2] / steps / 1e6:.1f}" for r in rows[:6])
        print(f"{str(q):>12} {n / steps:>8.0f} {tot / steps / 1e6:>9.2f}   {tops}")


def timeline(cur, t_from, steps):
    """One step (the middle one of the window) as per-stream segments: runs of kernels on one stream separated by less
    than 300 us are merged; prints start/end (ms into the step), busy ms, launches and the dominant kernel."""
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    qcol = next((c for c in ("stream_id", "queue_id") if c in cols), None)
    if qcol is None:
        return
    ks = cur.execute(f"select start, end, name, {qcol} from kernels where start >= ? order by start", (t_from,)).fetchall()
    marks = [k[0] for k in ks if "multi_tensor_apply" in k[2]]
    # step boundaries: the first optimiser kernel after a pause in optimiser kernels of nearly half a step
    span = (ks[-1][1] - ks[0][0]) / steps
    bounds = [marks[0]] + [b for a, b in zip(marks, marks[1:]) if b - a > 0.45 * span]
    if len(bounds) < 3:
        return
    mid = len(bounds) // 2
    a, b = bounds[mid - 1], bounds[mid]
    print(f"# timeline of one step ({(b - a) / 1e6:.1f} ms under the profiler): stream, from-to ms, busy ms, launches, top kernel")
    segs = {}
    out = []
    for s0, e0, name, q in ks:
        if s0 < a or s0 >= b:
            continue
        seg = segs.get(q)
        if seg is None or s0 - seg[1] > 3e5:
            if seg is not None:
                out.append((q, seg))
            seg = segs[q] = [s0, e0, 0, 0, {}]
        seg[1] = max(seg[1], e0)
        seg[2] += e0 - s0
        seg[3] += 1
        seg[4][name] = seg[4].get(name, 0) + e0 - s0
    out += list(segs.items())
    for q, seg in sorted(out, key=lambda t: t[1][0]):
        top = max(seg[4].items(), key=lambda kv: kv[1])[0].split("(")[0].split("::")[-1][:40]
        print(f"  stream {q}: {(seg[0] - a) / 1e6:8.2f} - {(seg[1] - a) / 1e6:8.2f}  busy {seg[2] / 1e6:7.2f}  n {seg[3]:5d}  {top}")


if __name__ == "__main__":
    main()
