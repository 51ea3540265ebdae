#!/usr/bin/env python3
"""MFMA-busy per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES pass:
  python tools/mfma_summary.py counter_collection.csv kernel_trace.csv
util = SQ_VALU_MFMA_BUSY_CYCLES / (dispatch duration x 2.4 GHz x 1024 SIMDs)   (MI355X_MICROARCH.md: the counter counts
cycles, one per SIMD-cycle the matrix pipe is busy; 256 CUs x 4 SIMDs; 2.4 GHz = the clock the peak figures assume).
of_busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 x 1024), when the pass collected SQ_BUSY_CYCLES: the same
numerator over the cycles the shader engines actually counted during the dispatch (32 SQ instances: 8 XCDs x 4 SEs;
for every chip-filling kernel of these runs SQ_BUSY_CYCLES / 32 / duration = 2.0 - 2.1 GHz: the clock under load)."""
import csv
import re
import statistics
import sys


def short(name):
    m = re.search(r"(k_[a-z0-9_]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:40]


def main():
    busy, dur, sq = {}, {}, {}
    disp = {}
    for r in csv.DictReader(open(sys.argv[2])):
        disp[r["Dispatch_Id"]] = (short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    for r in csv.DictReader(open(sys.argv[1])):
        d = disp.get(r["Dispatch_Id"])
        if d is None:
            continue
        if r["Counter_Name"] == "SQ_BUSY_CYCLES":
            sq.setdefault(d[0], []).append(float(r["Counter_Value"]))
        if r["Counter_Name"] != "SQ_VALU_MFMA_BUSY_CYCLES":
            continue
        busy.setdefault(d[0], []).append(float(r["Counter_Value"]))
        dur.setdefault(d[0], []).append(d[1])
    for k in sorted(busy, key=lambda k: -statistics.median(busy[k])):
        b, u = statistics.median(busy[k]), statistics.median(dur[k])
        if b <= 0:
            continue
        q = statistics.median(sq[k]) if sq.get(k) else 0.0
        extra = f"  of_busy {b / (q / 32 * 1024):.3f}  clock {q / 32 / u / 1e3:.2f} GHz" if q > 0 else ""
        print(f"{k:44s} launches {len(busy[k]):4d}  median {u:8.1f} us  MFMA busy cycles {b:14.0f}  util {b / (u * 1e-6 * 2.4e9 * 1024):.3f}{extra}")


if __name__ == "__main__":
    main()
