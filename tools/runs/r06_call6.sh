# round 6 call 6: whole GPU suite on the LDS-DMA GEMM tree, then the driver-style bench line (TTFT bs=1 / bs=32)
cd /root/repo
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06_gputest_call6.txt; cat gpurun_out/r06_gputest_call6.txt
python bench.py > gpurun_out/r06_bench_call6.json.log 2> gpurun_out/r06_bench_call6.err; tail -c 600 gpurun_out/r06_bench_call6.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r06_bench_call6.json.log').read().strip().splitlines()[-1])
print('value',j['value'],'ms/step',j['ms_per_step'],'ttft_p50',j.get('ttft_p50_ms'))
print('ttft',json.dumps(j.get('ttft'))[:600])
print('bs32',json.dumps({k:v for k,v in j.get('bs32',{}).items() if k!='roofline'})[:900])
print('roofline',j['roofline']['frac'],j['roofline']['us_per_launch'])
PY
