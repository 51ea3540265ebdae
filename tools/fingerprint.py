"""Fingerprint of the kernel sources (graph-hscn_amd/csrc/*.hip, *.h + include/hscn.h): what ties a committed
profile (PMC traffic, kernel stats) to the code it was collected from.  The GPU box has no .git, so a commit id
cannot be asked for there; a content hash can."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_fingerprint():
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(ROOT, "graph-hscn_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "graph-hscn_amd", "csrc", "*.h")) +
                   [os.path.join(ROOT, "include", "hscn.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


if __name__ == "__main__":
    print(csrc_fingerprint())
