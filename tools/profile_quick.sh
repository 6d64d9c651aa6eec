#!/bin/bash
# Quick counter profile of the bench (run on the GPU box via gpurun): kernel stats + the two SQ passes that tell issue-bound from
# waiting.  Usage: tools/profile_quick.sh <tag> [bench args] -> gpurun_out/prof_<tag>/ ; summarise with tools/summarize_prof.py
set -e
TAG=${1:-q}
shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline-baseline --feed resident $*"
# PMC_ONLY="counter list": just that one pass (e.g. the instruction-cache counters), no trace
if [ -n "$PMC_ONLY" ]; then
  rocprofv3 --kernel-trace --pmc $PMC_ONLY --output-format csv -d $OUT/pmcx -- python3 $ARGS > $OUT/pmcx.log 2>&1
  exit 0
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc$i -- python3 $ARGS > $OUT/pmc$i.log 2>&1 || echo "pmc pass $i failed" >> $OUT/errors.log
done
