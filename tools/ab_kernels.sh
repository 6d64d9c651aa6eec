#!/bin/bash
# GPU box: per-kernel times (rocprofv3 --kernel-trace --stats of the default bench step) of the in-tree library and of an alternative build.
# Usage: tools/ab_kernels.sh ALT.so
ALT=$1
L=bwa-mem-quickassist_amd/libbwamem_hip.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp $L /tmp/new.so
for v in tree alt; do
  [ $v = alt ] && cp $ALT $L
  rm -rf /tmp/ab_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$v -- python3 bench.py --steps 3 --warmup 1 --feed resident --no-cpu-baseline --no-pipeline-baseline > /dev/null 2>&1
done
cp /tmp/new.so $L
python3 - <<'PY'
import csv,glob
def load(v):
    f=glob.glob(f"/tmp/ab_{v}/*/*kernel_stats.csv")[0]
    return {r["Name"]:(int(r["Calls"]),float(r["TotalDurationNs"])/1e6) for r in csv.DictReader(open(f))}
a,b=load("tree"),load("alt")
print(f"{'kernel':70s} {'calls':>5s} {'tree ms':>9s} {'alt ms':>9s} ratio")
for k,(c,t) in sorted(a.items(), key=lambda kv:-kv[1][1])[:22]:
    if k in b: print(f"{k[:70]:70s} {c:5d} {t:9.1f} {b[k][1]:9.1f} {t/b[k][1]:.3f}")
PY
