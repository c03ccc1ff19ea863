#!/usr/bin/env python
"""Per-kernel HBM traffic from rocprofv3 PMC passes (csv output): averages FETCH_SIZE / WRITE_SIZE per dispatch.

usage: tools/pmc_traffic.py <dir-with-*counter_collection.csv> [...]   -> prints one line per kernel and a JSON dict.
Units (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts wide
coalesced reads at half their size -> bytes_read = 2 * 1024 * FETCH_SIZE.  WRITE_SIZE is uncalibrated (x1024 only)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void\s+)?([A-Za-z_0-9:]+)", name)
    return m.group(1) if m else name


def main():
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    a = acc[k][row["Counter_Name"]]
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
    out = {}
    for k, cs in sorted(acc.items(), key=lambda kv: -sum(v[0] for v in kv[1].values())):
        e = {}
        for c, (tot, n) in cs.items():
            e[c + "_avg_per_dispatch"] = tot / n
            e["dispatches"] = n
        rd = 2.0 * 1024.0 * e.get("FETCH_SIZE_avg_per_dispatch", 0.0)
        wr = 1024.0 * e.get("WRITE_SIZE_avg_per_dispatch", 0.0)
        e["hbm_read_bytes_corrected"] = rd
        e["hbm_write_bytes_uncalibrated"] = wr
        out[k] = e
        print(f"{k:40s} n={e['dispatches']:5d} read {rd / 1e6:10.2f} MB  write {wr / 1e6:10.2f} MB per dispatch")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
