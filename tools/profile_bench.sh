#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + PMC passes of the default bench.
# Usage: tools/profile_bench.sh <tag> [bench args]  -> gpurun_out/prof_<tag>/...
set -e
TAG=${1:-r01}
shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# the baselines start child processes and burn CPU: they are not what is profiled
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
# PMC passes (each in its own run; no tracing domains beside kernel-trace)
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" \
         "FETCH_SIZE" "WRITE_SIZE" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc$i -- python3 $ARGS > $OUT/pmc$i.log 2>&1 || echo "pmc pass $i failed" >> $OUT/errors.log
done
find $OUT -name "*.csv" > $OUT/files.txt
