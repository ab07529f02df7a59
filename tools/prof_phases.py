"""Per-phase GPU-busy breakdown from a rocprofv3 --kernel-trace --marker-trace run of bench.py with RECMV_TIMING=1
(the phases of recmv/loop.py are roctx ranges bracketed by device syncs).

    RECMV_TIMING=1 RECMV_PHASE_LOG=/tmp/phases.txt rocprofv3 --kernel-trace --marker-trace -d /tmp/p -o run -- \
        python bench.py --steps 10 --warmup 2 ...
    python tools/prof_phases.py /tmp/p /tmp/phases.txt [skip_first_n_occurrences] > profiles/<name>.txt
(the rocpd `regions` view names every roctx range "roctxThreadRangeA"; the range order is taken from the log)
"""
import collections
import glob
import sqlite3
import sys


def short(name):
    """Compact kernel label: recmv kernels by name (+ template args), torch kernels by their functor."""
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    if n.startswith("recmv::"):
        return n.split("(")[0][7:]
    for key in ("Functor", "functor", "_kernel_cuda", "Ops<"):
        i = n.find(key)
        if i >= 0:
            j = max(n.rfind("::", 0, i), n.rfind("<", 0, i), n.rfind(" ", 0, i))
            return "at:" + n[j + 1:i + len(key)].strip("<:, ")
    return n.split("<")[0].split("(")[0][-40:]


def main():
    path = sys.argv[1]
    names = [ln.strip() for ln in open(sys.argv[2]) if ln.strip()]
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    if not path.endswith(".db"):
        path = glob.glob(path + "/**/*.db", recursive=True)[0]
    cur = sqlite3.connect(path).cursor()
    regions = cur.execute("select name, start, end from regions order by start").fetchall()
    assert len(regions) == len(names), (len(regions), len(names))
    regions = [("recmv:" + n, r[1], r[2]) for n, r in zip(names, regions)]
    kernels = cur.execute("select name, start, end from kernels order by start").fetchall()
    seen = collections.Counter()
    agg = collections.OrderedDict()
    ki = 0
    for name, rs, re_ in regions:
        seen[name] += 1
        while ki < len(kernels) and kernels[ki][1] < rs:
            ki += 1
        kj = ki
        busy, n = 0, 0
        per = collections.Counter()
        pern = collections.Counter()
        while kj < len(kernels) and kernels[kj][1] <= re_:
            d = kernels[kj][2] - kernels[kj][1]
            busy += d
            n += 1
            per[short(kernels[kj][0])] += d
            pern[short(kernels[kj][0])] += 1
            kj += 1
        ki = kj
        if seen[name] <= skip:
            continue
        a = agg.setdefault(name, dict(count=0, wall=0, busy=0, launches=0, per=collections.Counter(), pern=collections.Counter()))
        a["count"] += 1
        a["wall"] += re_ - rs
        a["busy"] += busy
        a["launches"] += n
        a["per"].update(per)
        a["pern"].update(pern)
    print(f"# {path}  (first {skip} occurrence(s) of every phase skipped)")
    print(f"# {'phase':<22} {'n':>4} {'wall_ms/occ':>12} {'gpu_busy_ms/occ':>16} {'busy%':>6} {'launches/occ':>13}   top kernels (ms/occ)")
    for name, a in agg.items():
        c = a["count"]
        top = ", ".join(f"{k.strip()}={v / c / 1e6:.2f}" for k, v in a["per"].most_common(14))
        print(f"{name[6:]:<22} {c:>4d} {a['wall'] / c / 1e6:>12.2f} {a['busy'] / c / 1e6:>16.2f} "
              f"{100.0 * a['busy'] / max(a['wall'], 1):>6.1f} {a['launches'] / c:>13.0f}   {top}")
        if len(sys.argv) > 4 and sys.argv[4] == "counts":
            print("    launches/occ by kernel: " + ", ".join(f"{k.strip()} x{v / c:.0f}" for k, v in a["pern"].most_common(30)))


if __name__ == "__main__":
    main()
