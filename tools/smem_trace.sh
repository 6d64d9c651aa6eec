python3 tools/make_pipeline_input.py /tmp/pin 3200004 > /dev/null 2>&1
BMH_SMEM_TRACE=1 BMH_KSW_DROPIN=1 LD_PRELOAD=$PWD/bwa-mem-quickassist_amd/libbwamem_hip_dropin.so oracle/_ref/bwa mem -t 16 -b 32768 /tmp/pin/ref.fa /tmp/pin/r1.fq /tmp/pin/r2.fq 2>gpurun_out/smem_trace.txt >/dev/null
python3 - <<'PY'
import re, statistics as st
rows=[list(map(float,re.findall(r"([\d.]+) ms",l))) for l in open("gpurun_out/smem_trace.txt") if "bmh_smem_batch" in l]
print(len(rows)); rows=rows[32:]
for i,n in enumerate(["prepare","upload+kernel","download","reorder"]): print(n, round(st.mean(r[i] for r in rows),2), round(st.median(r[i] for r in rows),2))
PY
grep Processed gpurun_out/smem_trace.txt
