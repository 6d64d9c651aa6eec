#!/usr/bin/env python3
"""One hash over the kernel sources (bwa-mem-quickassist_amd/csrc/*.hip, *.h): profiles derived from a build (PMC traffic, instruction
mixes) carry it, and bench.py only quotes them when it matches the tree it runs from."""
import glob
import hashlib
import os


def csrc_sha(root=None):
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "bwa-mem-quickassist_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "bwa-mem-quickassist_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_sha())
