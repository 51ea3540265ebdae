#!/usr/bin/env python3
"""Per-kernel medians of the SQ counters collected by tools/run_sq_counters.sh.

  python tools/sq_summary.py sq_pass1.csv sq_pass2.csv

Units (MI355X_MICROARCH.md, cycle constants): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES count
quad-cycles summed over waves (or SQs), SQ_VALU_MFMA_BUSY_CYCLES counts cycles, SQ_INSTS_* count wave-instructions.
"""
import csv
import re
import statistics
import sys


def main():
    per = {}
    for path in sys.argv[1:]:
        for r in csv.DictReader(open(path)):
            m = re.search(r"\b(k_[a-z0-9_]+(?:<[^>]*>)?)", r["Kernel_Name"])
            if not m:
                continue
            per.setdefault(m.group(1), {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k in sorted(per):
        c = {n: statistics.median(v) for n, v in per[k].items()}
        n = min(len(v) for v in per[k].values())
        print(f"== {k}  (launches {n})")
        for name in sorted(c):
            print(f"   {name:28s} {c[name]:16.0f}")
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS"):
                if name in c:
                    print(f"   {name + ' / WAVE_CYCLES':28s} {c[name] / wc:16.3f}")
        if c.get("SQ_WAVES") and c.get("SQ_INSTS_VALU"):
            w = c["SQ_WAVES"]
            print(f"   per wave: VALU {c['SQ_INSTS_VALU'] / w:.0f}  SALU {c.get('SQ_INSTS_SALU', 0) / w:.0f}  "
                  f"LDS {c.get('SQ_INSTS_LDS', 0) / w:.0f}")


if __name__ == "__main__":
    main()
