#!/bin/bash
# GPU box: kernel trace + PMC passes of tools/sw_small_batches.py (sw_wave_kernel against sw_lane_kernel at small batch sizes).
# Usage: tools/profile_sw_wave.sh <tag>   -> gpurun_out/prof_<tag>/
set -e
TAG=${1:-r02_sw_wave}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/sw_small_batches.py > $OUT/trace.log 2>&1
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
         "FETCH_SIZE" "WRITE_SIZE" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc$i -- python3 tools/sw_small_batches.py > $OUT/pmc$i.log 2>&1 || echo "pmc pass $i failed" >> $OUT/errors.log
done
