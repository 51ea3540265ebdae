"""Fingerprint of the sources of the kernels whose PMC traffic is committed (profiles/rNN_pmc_traffic.json: the
graph-resident step kernels and the streaming SpMM) -- graph-hscn_amd/csrc/resident*.h, resident*.hip, spmm.hip,
hscn_common.h: what ties that profile to the code it was collected from.  The GPU box has no .git, so a commit id
cannot be asked for there; a content hash can.  (Files of other kernels -- dense.hip, linear.hip, structure.hip ... --
are left out on purpose: a change there does not make the step's traffic figure stale.)"""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_fingerprint():
    h = hashlib.sha1()
    d = os.path.join(ROOT, "graph-hscn_amd", "csrc")
    files = sorted(glob.glob(os.path.join(d, "resident*.hip")) + glob.glob(os.path.join(d, "resident*.h")) +
                   [os.path.join(d, "spmm.hip"), os.path.join(d, "hscn_common.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


if __name__ == "__main__":
    print(csrc_fingerprint())
