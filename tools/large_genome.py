#!/usr/bin/env python3
"""One whole-pipeline run on a genome whose doubled coordinates leave 32 bits (2 * l_pac > 2^31): the reference's `bwa mem` and the same
binary with the library preloaded, SAM compared, reads/s of both, index build time.  The index is built on the spot by the compiled
reference's `bwa index` (BWT-SW; about 0.5 us per base on the GPU boxes' hosts) -- run it under gpurun with the longest timeout:
    python tools/large_genome.py [genome_bp=1100000000] [reads=800000]
Progress goes to gpurun_out/large_genome_index.log while the index is built."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench

genome = int(sys.argv[1]) if len(sys.argv) > 1 else 1_100_000_000
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 800_000
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
ncores, note = bench.host_cores()
t0 = time.time()


def _heartbeat():  # the index build is silent for minutes at a time ("Construct SA from BWT and Occ"): keep a file under gpurun_out/ moving
    import threading

    def beat():
        while True:
            time.sleep(45)
            with open(os.path.join(ROOT, "gpurun_out", "large_genome_heartbeat.log"), "a") as f:
                f.write(f"{time.time() - t0:.0f} s\n")
    threading.Thread(target=beat, daemon=True).start()


_heartbeat()
r = bench.pipeline_baseline(reads, genome, ncores, batch=32768, index_log=os.path.join(ROOT, "gpurun_out", "large_genome_index.log"))
r["total_s"] = time.time() - t0
r["cores_note"] = note
r["doubled_coordinates_exceed_2^31"] = 2 * genome > 2 ** 31
print(json.dumps(r))
