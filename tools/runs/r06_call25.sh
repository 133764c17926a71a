#!/bin/bash
# round 6, GPU call 25: whole GPU suite on the tree with the preloaded strip entry point (FULL instances), then the driver-style bench line
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r06_gputest_call25.txt; cat gpurun_out/r06_gputest_call25.txt
python bench.py > gpurun_out/r06_bench_call25.json.log 2> gpurun_out/r06_bench_call25.err; tail -c 600 gpurun_out/r06_bench_call25.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r06_bench_call25.json.log').read().strip().splitlines()[-1])
print('value',j['value'],'ms/step',j['ms_per_step'],'ttft_p50',j.get('ttft_p50_ms'))
print('ttft',json.dumps(j.get('ttft'))[:600])
print('bs32',json.dumps({k:v for k,v in j.get('bs32',{}).items() if k!='roofline'})[:900])
print('bs32 roofline',json.dumps(j.get('bs32',{}).get('roofline'))[:400])
print('bs128',json.dumps(j.get('bs128'))[:900])
print('roofline',j['roofline']['frac'],j['roofline']['us_per_launch'])
PY
