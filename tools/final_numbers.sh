#!/bin/bash
# The round's measurement set (run on the GPU box via gpurun): the default bench line (resident + host-fed steps, both CPU baselines),
# one stream, the two other read shapes, round 1's flat batch, and the two-rank rehearsal on one device.  Files under gpurun_out/.
O=gpurun_out
python bench.py > $O/r03_final_default.json 2> $O/r03_final_default.err
tail -2 $O/r03_final_default.err
python bench.py --streams 1 --no-pipeline-baseline --no-cpu-baseline --feed resident > $O/r03_final_s1.json 2>/dev/null
python bench.py --shape mixed100-300 --no-pipeline-baseline --no-cpu-baseline --feed resident > $O/r03_final_mixed.json 2>/dev/null
python bench.py --shape 250bp --no-pipeline-baseline --no-cpu-baseline --feed resident > $O/r03_final_250.json 2>/dev/null
python bench.py --workload se1m --no-cpu-baseline > $O/r03_final_se1m.json 2>/dev/null
python bench.py --gpus 2 --oversubscribe --no-pipeline-baseline --no-cpu-baseline --pairs 4000000 > $O/r03_final_2ranks.json 2>/dev/null
python - <<PY
import json
for f in ("default","s1","mixed","250","se1m","2ranks"):
    try:
        d=json.load(open("gpurun_out/r03_final_%s.json"%f))
        print(f, round(d["value"]/1e6,2), round(d["ms_per_step"],1), d.get("value_streamed") and round(d["value_streamed"]/1e6,1), d.get("stages_ms_per_step"), d.get("parity","")[:12], d.get("n_gpus"), d.get("per_rank_ms_per_step"))
    except Exception as e: print(f, "ERR", e)
d=json.load(open("gpurun_out/r03_final_default.json")); r=d["roofline"]
print({k:r[k] for k in ("achieved","frac","traffic","kernel","kernel_ms")}); v=r["valu_issue"]; print({k:v.get(k) for k in ("achieved","peak_mix_weighted","frac","cycles_per_instruction_mix_weighted","valu_insts_per_launch")})
p=d["cpu_baseline_pipeline"]; print(p["value"],p["dut_value"],p["dut_over_ref"],p["sam_identical"],p["dut_value_after_first_chunk"],p["ref_cpu_s"],p["dut_cpu_s"],p["dut_detail"]["chunk_real_s"]); print(d["cpu_baseline"]["value"], d["stage_rates"]); s=d["streamed"]; print({k:s[k] for k in ("ms_per_step","h2d_bytes_per_step","d2h_bytes_per_step","h2d_GBps_copy_stream")})
for k in r["kernels"]: print(k["kernel"][:50], round(k["ms"],1))
PY
