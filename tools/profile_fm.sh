#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + PMC passes of the FM-index (seeding) measurement.
# Usage: tools/profile_fm.sh <tag>  -> gpurun_out/prof_<tag>/...
set -e
TAG=${1:-fm}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --global-tasks 0 --sw-tasks 0 --reads 20000"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc$i -- python3 $ARGS > $OUT/pmc$i.log 2>&1 || echo "pmc pass $i failed" >> $OUT/errors.log
done
find $OUT -name "*.csv" > $OUT/files.txt
