#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace of the whole pipeline -- the reference's `bwa mem` with the
# library preloaded -- to see how busy the GPU is during a chunk and which kernels take the time.
# Usage: tools/profile_pipeline.sh <tag> [n_reads] [threads]
set -e
TAG=${1:-r02_pipeline}; N=${2:-1600000}; T=${3:-16}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 tools/make_pipeline_input.py /tmp/bmh_pin $N > $OUT/input.log 2>&1
export BMH_VERBOSE=1
# (the shim is handed to the profiled program only: rocprofv3's own launcher must not load it)
rocprofv3 --preload $PWD/bwa-mem-quickassist_amd/libbwamem_hip_dropin.so --kernel-trace --stats --output-format csv -d $OUT/trace -- $PWD/oracle/_ref/bwa mem -t $T -b 32768 /tmp/bmh_pin/ref.fa /tmp/bmh_pin/r1.fq /tmp/bmh_pin/r2.fq > /tmp/bmh_pin/dut.sam 2> $OUT/dut.err
grep -E "chunk of|Processed|thread-seconds|phase 1 so far" $OUT/dut.err > $OUT/summary.txt || true
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
