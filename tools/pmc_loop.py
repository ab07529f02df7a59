"""HBM traffic per launch of the loop's MFMA kernels, from rocprofv3 PMC passes over bench.py itself.

    # on the GPU box, one pass per counter (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass)
    cd /tmp && export TMPDIR=/tmp
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o run -- python $REPO/bench.py --steps 3 --warmup 1 \
          --settle-iters 40 --no-cpu-baseline --no-mc --no-alt-mode --no-hbm-kernels --no-kernel-events
    done
    python tools/pmc_loop.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE --measured-at "<commit sha>, <date>" > profiles/rNN_pmc_loop.json

Per kernel name (template arguments kept, parameter list dropped): launches seen, mean FETCH_SIZE / WRITE_SIZE per
launch in bytes (rocprofv3 reports KiB-like units; FETCH_SIZE doubled: on gfx950 a wide coalesced streaming read is
tallied at half its size; WRITE_SIZE is uncalibrated and taken as reported) and their sum = `traffic`.  The means are
over ALL launches of the kernel in the run (settle + timed steps): the loop's launch mix, not one shape.
"""
import csv
import glob
import json
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "").replace("recmv::(anonymous namespace)::", "")
    depth, out = 0, []
    for ch in name:                      # cut the parameter list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()


def collect(root):
    acc = defaultdict(list)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if "recmv::" in row.get("Kernel_Name", ""):
                    acc[(short(row["Kernel_Name"]), row["Counter_Name"])].append((int(row.get("Dispatch_Id", 0) or 0),
                                                                                  float(row["Counter_Value"])))
    return {k: [v for _, v in sorted(rows)] for k, rows in acc.items()}      # dispatch order: the two passes launch the same sequence


def main():
    acc = {}
    argv = sys.argv[1:]
    measured_at = None
    if "--measured-at" in argv:
        i = argv.index("--measured-at")
        measured_at = argv[i + 1]
        argv = argv[:i] + argv[i + 2:]
    for root in argv:
        acc.update(collect(root))
    kernels = {}
    for (name, ctr), vals in acc.items():
        k = kernels.setdefault(name, {"launches": len(vals)})
        mean = sum(vals) / len(vals) * 1024.0
        if ctr == "FETCH_SIZE":
            k["fetch_bytes_per_launch"] = round(2.0 * mean)
        elif ctr == "WRITE_SIZE":
            k["write_bytes_per_launch"] = round(mean)
    for k in kernels.values():
        if "fetch_bytes_per_launch" in k and "write_bytes_per_launch" in k:
            k["traffic_bytes_per_launch"] = k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"]
    # The loop launches a kernel on very different shapes (a dW product over 3 k rays and over 250 k vertices): bench.py's roofline
    # object brackets only the launches above 4 GFLOP, so the matching traffic figure is the mean over the LARGE launches — those
    # whose fetch is at least a quarter of the kernel's largest (both passes see the same launch sequence, so the k-th launch of
    # the WRITE pass is the k-th of the FETCH pass).
    for name in kernels:
        f, w = acc.get((name, "FETCH_SIZE")), acc.get((name, "WRITE_SIZE"))
        if not f or not w or len(f) != len(w):
            continue
        cut = 0.25 * max(f)
        idx = [i for i, v in enumerate(f) if v >= cut]
        if idx and len(idx) < len(f):
            fm = sum(f[i] for i in idx) / len(idx) * 1024.0 * 2.0
            wm = sum(w[i] for i in idx) / len(idx) * 1024.0
            kernels[name]["large_launches"] = {"launches": len(idx), "fetch_bytes_per_launch": round(fm), "write_bytes_per_launch": round(wm),
                                               "traffic_bytes_per_launch": round(fm + wm),
                                               "rule": "launches whose FETCH_SIZE is >= 25 % of the kernel's largest"}
    top = dict(sorted(kernels.items(), key=lambda kv: -kv[1].get("traffic_bytes_per_launch", 0) * kv[1]["launches"])[:40])
    json.dump({"measured_at": measured_at, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over bench.py (tools/pmc_loop.py); FETCH_SIZE x2 "
                         "(gfx950 wide-read correction), WRITE_SIZE as reported", "kernels": top}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
