#!/usr/bin/env python3
"""Resolve and summarise a tools/cpuprof.c dump: python tools/cpuprof_report.py /tmp/prof.txt [top]
Prints CPU share by leaf function, by the innermost frame inside this repository's libraries ("own"), and by (own, leaf) pair.
Static functions are resolved with addr2line (the libraries are built with -g)."""
import os
import subprocess
import sys
from collections import Counter

path, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
samples = []
need = {}
for line in open(path):
    if line.startswith("#") or not line.strip():
        continue
    fr = []
    for part in line.strip().split(";"):
        if not part:
            continue
        obj, off, sym = part.split(" ", 2)
        fr.append((obj, off, sym))
        if "bwamem_hip" in obj or obj.endswith("/bwa"):
            need.setdefault(obj, set()).add(off)
    samples.append(fr)
names = {}
for obj, offs in need.items():
    offs = sorted(offs)
    main_exe = obj.endswith("/bwa")
    for i in range(0, len(offs), 500):
        chunk = offs[i:i + 500]
        out = subprocess.run(["addr2line", "-f", "-e", obj] + ["0x" + o for o in chunk], capture_output=True, text=True).stdout.splitlines()
        for o, fn in zip(chunk, out[0::2]):
            names[(obj, o)] = fn
def name(fr):
    obj, off, sym = fr
    n = names.get((obj, off))
    if n and n != "??":
        return n
    return (sym if sym != "?" else off) + "@" + os.path.basename(obj)
leaf, own, pair = Counter(), Counter(), Counter()
for fr in samples:
    if not fr:
        continue
    l = name(fr[0])
    o = next((name(f) for f in fr if "bwamem_hip" in f[0] or f[0].endswith("/bwa")), "(none)")
    leaf[l] += 1
    own[o] += 1
    pair[(o, l)] += 1
n = len(samples)
print(f"{n} samples = {n / 1000:.2f} CPU-seconds\n\n## by leaf")
for k, v in leaf.most_common(top):
    print(f"{100 * v / n:6.2f} %  {k}")
print("\n## by innermost frame in our libraries / the host program")
for k, v in own.most_common(top):
    print(f"{100 * v / n:6.2f} %  {k}")
print("\n## (own, leaf)")
for (o, l), v in pair.most_common(top):
    print(f"{100 * v / n:6.2f} %  {o}  <-  {l}")
