#!/bin/bash
# GPU box: same-box A/B of two builds of the library -- the in-tree one against bwa-mem-quickassist_amd/build/lib_alt.so -- on the default bench step.
# Usage: tools/ab_lib.sh [ALT.so] [bench args]
ALT=${1:-bwa-mem-quickassist_amd/build/lib_alt.so}; shift
L=bwa-mem-quickassist_amd/libbwamem_hip.so
cp $L /tmp/new.so
run() { python3 bench.py --feed resident --no-cpu-baseline --no-pipeline-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$TAG', round(d['ms_per_step'],2))"; }
for k in 1 2 3; do
cp /tmp/new.so $L; TAG=tree run "$@"
cp $ALT $L; TAG=alt run "$@"
done
cp /tmp/new.so $L
