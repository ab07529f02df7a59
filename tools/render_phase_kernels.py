"""The render phase of an iteration on the device — from the root finder's last step to the optimiser step — out of a rocprofv3 kernel
trace of `bench.py --steps 6` (command: see tools/tail_timeline.py): per stream the launches, their summed run time, and the kernels
with the most time.  Under the tracer the HOST paces the launches (12 us each), so read the kernel-time sums, not the wall time.
    python tools/render_phase_kernels.py /tmp/tl"""
import collections
import glob
import re
import sqlite3
import sys

path = sys.argv[1]
if not path.endswith(".db"):
    path = glob.glob(path + "/**/*.db", recursive=True)[0]
con = sqlite3.connect(path)
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
qcol = "stream_id" if "stream_id" in cols else "queue_id"
rows = con.execute("select name, start, end, %s from kernels order by start" % qcol).fetchall()
adam = [r for r in rows if "multi_tensor_apply" in r[0] and "TensorListScalarListMetadata<float, 3>" in r[0]]
steps = []
for r in adam:
    if not steps or r[1] - steps[-1][-1][2] > 20e6:
        steps.append([r])
    else:
        steps[-1].append(r)


def short(name):
    s = name.replace("void ", "").replace("recmv::(anonymous namespace)::", "").replace("at::native::(anonymous namespace)::", "")
    s = s.replace("at::native::", "")
    return re.sub(r"\(.*", "", s)[:64]


for a, b in list(zip(steps[:-1], steps[1:]))[-2:]:
    t0, t1 = a[-1][2], b[0][1]
    it = [r for r in rows if r[1] >= t0 and r[2] <= t1]
    rf = [r for r in it if "rootfind_step_kernel" in r[0]]
    if not rf:
        continue
    tr = max(r[2] for r in rf)
    ph = [r for r in it if r[1] >= tr]
    print("iteration %.1f ms under the tracer; root finder ends at %.1f ms; render phase + tail: %.1f ms, %d launches, %.2f ms of kernel time"
          % ((t1 - t0) / 1e6, (tr - t0) / 1e6, (t1 - tr) / 1e6, len(ph), sum(r[2] - r[1] for r in ph) / 1e6))
    by_stream = collections.defaultdict(list)
    for r in ph:
        by_stream[r[3]].append(r)
    for q, rs in sorted(by_stream.items(), key=lambda kv: -sum(r[2] - r[1] for r in kv[1])):
        names = collections.defaultdict(lambda: [0, 0.0])
        for r in rs:
            names[short(r[0])][0] += 1
            names[short(r[0])][1] += (r[2] - r[1]) / 1e3
        print("  stream %s: %d launches, %.2f ms of kernel time (first at %.1f ms, last ends %.1f ms)"
              % (q, len(rs), sum(r[2] - r[1] for r in rs) / 1e6, (rs[0][1] - t0) / 1e6, (max(r[2] for r in rs) - t0) / 1e6))
        for k, v in sorted(names.items(), key=lambda kv: -kv[1][1])[:10]:
            print("     %5d x  %8.1f us  %s" % (v[0], v[1], k))
