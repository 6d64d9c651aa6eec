#!/usr/bin/env python3
"""Where do the host threads of the preload-shim pipeline spend a chunk -- on the CPU or waiting?  Runs the DUT
(`oracle/_ref/bwa mem` with the library preloaded) twice on the input of tools/make_pipeline_input.py: once with the
stage clocks reading wall time, once (BMH_TRACE_CPU=1) reading each thread's CPU time, and prints the stage sums and the
mean per-call figures of the BMH_DRIVER_TRACE lines side by side.  wall - cpu of a stage = time its threads were
blocked (GPU waits that sleep, the device gate, page faults).
Usage (GPU box): python tools/stage_clocks.py DIR [threads] [batch]      (DIR from tools/make_pipeline_input.py)"""
import os
import re
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d, t, b = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "16", sys.argv[3] if len(sys.argv) > 3 else "32768"
bwa = os.path.join(ROOT, "oracle", "_ref", "bwa")
lib = os.path.join(ROOT, "bwa-mem-quickassist_amd", "libbwamem_hip_dropin.so")


def run(cpu):
    env = dict(os.environ, LD_PRELOAD=lib, BMH_KSW_DROPIN="1", BMH_VERBOSE="1", BMH_DRIVER_TRACE="1")
    if cpu:
        env["BMH_TRACE_CPU"] = "1"
    p = subprocess.run([bwa, "mem", "-t", t, "-b", b, os.path.join(d, "ref.fa"), os.path.join(d, "r1.fq"), os.path.join(d, "r2.fq")],
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env, timeout=600)
    err = p.stderr.decode(errors="replace").splitlines()
    sums = [l for l in err if "thread-seconds so far" in l][-3:]
    proc = [l for l in err if "Processed" in l]
    calls = defaultdict(lambda: defaultdict(list))
    for l in err:
        m = re.match(r"\[bwamem_hip\] (bmh_\w+) ", l)
        if not m:
            continue
        for name, ms in re.findall(r"([A-Za-z+/ ().:_2]+?) (\d+\.\d+) ms", l):
            calls[m.group(1)][name.strip(" ,:;")].append(float(ms))
    return sums, proc, calls


run(False)  # warm the runtime
w = run(False)
c = run(True)
print("## stage sums over the run (thread-seconds): wall, then CPU\n")
for a, bb in zip(w[0], c[0]):
    print("wall:", a.split("] ", 1)[1])
    print("cpu :", bb.split("] ", 1)[1])
print("\n## per chunk\n")
for l in w[1]:
    print("wall run:", l)
for l in c[1]:
    print("cpu  run:", l)
print("\n## BMH_DRIVER_TRACE lines, mean ms per call (wall | cpu), calls\n")
for k in w[2]:
    for name in w[2][k]:
        a, bb = w[2][k][name], c[2][k].get(name, [0.0])
        print(f"{k:24s} {name[:60]:60s} {sum(a) / len(a):8.2f} | {sum(bb) / len(bb):8.2f}   x{len(a)}")
