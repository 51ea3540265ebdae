#!/usr/bin/env python3
"""HBM traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in
separate runs with --kernel-trace only, as MI355X_MICROARCH.md prescribes).

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT_F -- python3 bench.py --mode eager ...
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d OUT_W -- python3 bench.py --mode eager ...
  python tools/pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv out.json [commit]

traffic_bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (medians over launches): on gfx950 FETCH_SIZE
tallies 64 B per 128-B request of a 16 B/lane read, WRITE_SIZE is exact (both in KiB).  The JSON records the commit
(given on the command line: the GPU box has no .git) and the fingerprint of the kernel sources it was collected from;
bench.py marks the figure stale when the sources have changed since.
"""
import csv
import json
import os
import re
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fingerprint import csrc_fingerprint


def per_kernel(path, counter):
    vals = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"\b(k_[a-z0-9_]+)(<[^>]*>)?", r["Kernel_Name"])
        if m:
            vals.setdefault(m.group(1), []).append(float(r["Counter_Value"]))
            if m.group(2):
                vals.setdefault(m.group(1) + m.group(2).replace(" ", ""), []).append(float(r["Counter_Value"]))
    return vals


def main():
    fpath, wpath, out = sys.argv[1], sys.argv[2], sys.argv[3]
    commit = sys.argv[4] if len(sys.argv) > 4 else os.environ.get("HSCN_COMMIT", "unrecorded")
    f = per_kernel(fpath, "FETCH_SIZE")
    w = per_kernel(wpath, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) & set(w)):
        fm, wm = statistics.median(f[k]), statistics.median(w[k])
        res[k] = {"fetch_size_kb": fm, "write_size_kb": wm, "traffic_bytes": int(2 * fm * 1024 + wm * 1024),
                  "launches": min(len(f[k]), len(w[k]))}
        print(f"{k:44s} fetch {fm:10.1f} KiB  write {wm:10.1f} KiB  traffic {res[k]['traffic_bytes'] / 1e6:8.2f} MB  n={res[k]['launches']}")
    # the streaming SpMM at its scaled shape (bench.py's streaming_spmm_scaled leg: 4 096 graphs, H = 128): the
    # k_spmm<4,0,2> launches of the run
    for k in list(res):
        if k.startswith("k_spmm<4,0,2>"):
            res["k_spmm_scaled_H128"] = res[k]
    doc = {"_method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (each with --kernel-trace only), "
                      "median over launches; traffic_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024: on gfx950 FETCH_SIZE counts "
                      "64 B per 128-B request for 16 B/lane streaming reads (MI355X_MICROARCH.md, HBM section), WRITE_SIZE "
                      "is exact for 16 B/lane stores",
           "_commands": ["tools/run_profiles.sh <tag> (through gpurun)"],
           "_commit": commit, "_csrc_sha": csrc_fingerprint(), "kernels": res}
    json.dump(doc, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
