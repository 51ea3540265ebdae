#!/usr/bin/env python3
"""Per-kernel resource table from `make -C graph-hscn_amd asm` (-Rpass-analysis=kernel-resource-usage):
  python tools/resource_table.py [graph-hscn_amd/build/asm] > profiles/rNN_kernel_resources.txt
columns: VGPRs, AGPRs, SGPRs, spilled SGPRs / VGPRs, scratch bytes/lane, occupancy (waves/SIMD), LDS bytes (static)."""
import glob
import os
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "-n"], input="\n".join(names), capture_output=True, text=True).stdout
        return out.splitlines()
    except OSError:
        return names


def main():
    d = sys.argv[1] if len(sys.argv) > 1 else "graph-hscn_amd/build/asm"
    rows = []
    for f in sorted(glob.glob(os.path.join(d, "*.usage.txt"))):
        cur = None
        for ln in open(f):
            m = re.search(r"Function Name: (\S+)", ln)
            if m:
                cur = {"file": os.path.basename(f).replace(".usage.txt", ""), "name": m.group(1)}
                rows.append(cur)
                continue
            if cur is None:
                continue
            for key, pat in (("vgpr", r"VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"SGPRs: (\d+)"),
                             ("sspill", r"SGPRs Spill: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"),
                             ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                             ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
                m = re.search(pat, ln)
                if m and key not in cur:
                    cur[key] = int(m.group(1))
    names = demangle([r["name"] for r in rows])
    print(f"{'file':14s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'sSpill':>6s} {'vSpill':>6s} {'scratch':>7s} {'occ':>3s} {'lds':>6s}  kernel")
    for r, n in zip(rows, names):
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"\(.*\)$", "", n)
        print(f"{r['file']:14s} {r.get('vgpr', 0):5d} {r.get('agpr', 0):5d} {r.get('sgpr', 0):5d} {r.get('sspill', 0):6d} "
              f"{r.get('vspill', 0):6d} {r.get('scratch', 0):7d} {r.get('occ', 0):3d} {r.get('lds', 0):6d}  {n}")


if __name__ == "__main__":
    main()
