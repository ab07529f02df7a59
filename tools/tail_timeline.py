"""One iteration's kernels by HIP stream, from a rocprofv3 kernel trace of `bench.py --steps 6`: per 5 ms window of the LAST timed
iteration, for every stream (queue) the number of kernels, the time they ran and the largest kernel names — who holds the device
when, and whether the iteration's tail (after the large products end) is kernel time or gaps between dependent launches.

    cd /tmp && rocprofv3 --kernel-trace -d /tmp/tl -o run -- python $REPO/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-mc \
        --no-serial-pass --no-hbm-kernels --no-config2 --no-kernel-events
    python tools/tail_timeline.py /tmp/tl"""
import collections
import glob
import sqlite3
import sys

path = sys.argv[1]
if not path.endswith(".db"):
    path = glob.glob(path + "/**/*.db", recursive=True)[0]
con = sqlite3.connect(path)
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
rows = con.execute("select name, start, end%s from kernels order by start" % ((", " + qcol) if qcol else ", 0")).fetchall()
# iteration boundaries: Adam's multi-tensor kernels (one burst per optimiser step; the curve AdamW and SGD bursts are much smaller)
adam = [r for r in rows if "multi_tensor_apply" in r[0] and "TensorListScalarListMetadata<float, 3>" in r[0]]      # Adam's addcdiv
bursts = []
for r in adam:
    if not bursts or r[1] - bursts[-1][1] > 20e6:
        bursts.append([r[1], r[2]])
    else:
        bursts[-1][1] = r[2]
# keep the bursts that end an iteration of the main Adam (the longest ones)
if len(bursts) < 3:
    print("could not find the optimiser steps (%d bursts); columns: %s" % (len(bursts), cols))
    sys.exit(0)
# the iteration (end of Adam to end of Adam) with the most kernels: a full three-frame batch (an epoch's last position has one frame)
best, cnt = None, -1
for a, b in zip(bursts[:-1], bursts[1:]):
    n = sum(1 for r in rows if a[1] <= r[1] and r[2] <= b[1])
    if n > cnt:
        best, cnt = (a[1], b[1]), n
t0, t1 = best
print("# iteration with the most kernels (%d): %.2f ms; streams by '%s'" % (cnt, (t1 - t0) / 1e6, qcol))
it = [r for r in rows if r[1] >= t0 and r[2] <= t1]
W = 5e6
nwin = int((t1 - t0) // W) + 1
by = collections.defaultdict(lambda: [[0, 0.0, collections.Counter()] for _ in range(nwin)])
for name, s, e, q in it:
    w = int((s - t0) // W)
    c = by[q][w]
    c[0] += 1
    c[1] += (e - s) / 1e6
    short = name.replace("void ", "").replace("recmv::(anonymous namespace)::", "").replace("at::native::(anonymous namespace)::", "")
    short = short.replace("at::native::", "").split("<")[0].split("(")[0][:48]
    c[2][short] += (e - s) / 1e6
tot = {q: sum(c[1] for c in ws) for q, ws in by.items()}
for q in sorted(by, key=lambda q: -tot[q]):
    print("stream %s: %d kernels, %.1f ms of kernel time" % (q, sum(c[0] for c in by[q]), tot[q]))
    for w, (n, ms, names) in enumerate(by[q]):
        if n:
            top = ", ".join("%s %.1f" % (k, v) for k, v in names.most_common(3))
            print("   %3d-%3d ms: %4d kernels  %5.2f ms busy   %s" % (w * 5, w * 5 + 5, n, ms, top))
