"""Summarise a rocprofv3 --pmc run (csv output): per kernel name and counter, dispatch count and the mean /
min / max counter value per dispatch.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc -o run -- <cmd>
    python tools/pmc_summary.py /tmp/pmc [name-substring ...] > profiles/<name>.txt

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB-like units of the TCC request counters; the raw value
is printed together with `bytes = value * 1024` and, for FETCH_SIZE, the gfx950 correction x2
(MI355X_MICROARCH.md, HBM section: wide coalesced streaming reads are tallied at half their size).
"""
import csv
import glob
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    filters = sys.argv[2:]
    files = glob.glob(root + "/**/*counter_collection.csv", recursive=True)
    if not files:
        print("no *counter_collection.csv under", root)
        return
    acc = defaultdict(list)
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                if filters and not any(s in name for s in filters):
                    continue
                acc[(name, row.get("Counter_Name", "?"), row.get("Grid_Size", "?"))].append(float(row["Counter_Value"]))
    print(f"# {', '.join(files)}")
    print(f"# {'kernel':<70} {'grid':>10} {'counter':<28} {'n':>6} {'mean':>16} {'min':>16} {'max':>16}  notes")
    for (name, ctr, grid), vals in sorted(acc.items(), key=lambda kv: (kv[0][0], kv[0][2], kv[0][1])):
        mean = sum(vals) / len(vals)
        note = ""
        if ctr == "FETCH_SIZE":
            note = f"bytes={mean * 1024:.4g}  x2(gfx950 wide-read correction)={2 * mean * 1024:.4g}"
        elif ctr == "WRITE_SIZE":
            note = f"bytes={mean * 1024:.4g} (uncalibrated)"
        print(f"{name[:70]:<70} {grid:>10} {ctr:<28} {len(vals):>6d} {mean:>16.6g} {min(vals):>16.6g} {max(vals):>16.6g}  {note}")


if __name__ == "__main__":
    main()
