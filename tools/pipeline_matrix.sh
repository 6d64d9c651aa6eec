#!/bin/bash
# GPU box: the DUT (bwa mem + preload shim) over a list of environment settings, reads/s from the Processed lines.
# Usage: tools/pipeline_matrix.sh DIR "-t 16 -b 32768" "VAR=val VAR=val" "VAR=val" ...   (DIR from tools/make_pipeline_input.py)
DIR=$1; ARGS=$2; shift 2
L=$PWD/bwa-mem-quickassist_amd/libbwamem_hip_dropin.so
run() { # env-string -> prints reads/s
  env $1 BMH_KSW_DROPIN=1 BMH_VERBOSE=1 LD_PRELOAD=$L oracle/_ref/bwa mem $ARGS $DIR/ref.fa $DIR/r1.fq $DIR/r2.fq 2> /tmp/m.err | md5sum > /tmp/m.md5
  python3 - "$1" <<'PY'
import re,sys
t=open("/tmp/m.err").read()
m=re.findall(r"Processed (\d+) reads in ([\d.]+) CPU sec, ([\d.]+) real sec",t)
n=sum(int(a) for a,_,_ in m); r=sum(float(c) for _,_,c in m); cpu=sum(float(b) for _,b,_ in m)
# steady state: leave the first chunk out
n2=sum(int(a) for a,_,_ in m[1:]); r2=sum(float(c) for _,_,c in m[1:])
print(f"{sys.argv[1]:60s} {n/r/1e6:6.3f} M reads/s  (after chunk 1: {n2/max(r2,1e-9)/1e6:6.3f})  chunks {' '.join(c for _,_,c in m)}  cpu {cpu:6.2f} s  md5 {open('/tmp/m.md5').read()[:8]}", flush=True)
PY
}
run "BMH_NOP=1" > /dev/null   # warm the runtime
for e in "$@"; do sleep 3; run "$e"; done   # (a pause: the previous process is still tearing its GPU state down)
