#!/usr/bin/env python3
"""HBM traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in
separate runs with --kernel-trace only, as MI355X_MICROARCH.md prescribes).

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT_F -- python3 bench.py --mode eager ...
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d OUT_W -- python3 bench.py --mode eager ...
  python tools/pmc_traffic.py OUT_F/*/*counter_collection.csv OUT_W/*/*counter_collection.csv [json to update]

traffic_bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (medians over launches): on gfx950 FETCH_SIZE
tallies 64 B per 128-B request of a 16 B/lane read, WRITE_SIZE is exact (both in KiB).
"""
import csv
import re
import json
import statistics
import sys


def per_kernel(path, counter):
    vals = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"\b(k_[a-z0-9_]+)", r["Kernel_Name"])
        if m:
            vals.setdefault(m.group(1), []).append(float(r["Counter_Value"]))
    return vals


def main():
    fpath, wpath = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else None
    f = per_kernel(fpath, "FETCH_SIZE")
    w = per_kernel(wpath, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) & set(w)):
        if not k.startswith("k_"):
            continue
        fm, wm = statistics.median(f[k]), statistics.median(w[k])
        res[k] = {"fetch_size_kb": fm, "write_size_kb": wm, "traffic_bytes": int(2 * fm * 1024 + wm * 1024),
                  "launches": min(len(f[k]), len(w[k]))}
        print(f"{k:28s} fetch {fm:10.1f} KiB  write {wm:10.1f} KiB  traffic {res[k]['traffic_bytes'] / 1e6:8.2f} MB  n={res[k]['launches']}")
    if out:
        doc = json.load(open(out))
        doc["kernels"].update(res)
        json.dump(doc, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
