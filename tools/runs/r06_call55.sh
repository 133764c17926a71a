#!/bin/bash
# round 6, GPU call 55 (records of the final tree: every node of the time-to-first-token path on preloaded kernel arguments): whole GPU suite, smoke(), the driver-style bench line, the rocprofv3
# --kernel-trace --stats summary of the bench command, kernel tables of the TTFT path (1 / 32 utterances) and of the 32- / 128-utterance step
R=$GRAFT_REPO_ROOT
cd $R || exit 1
export TMPDIR=/tmp
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | tail -12 > gpurun_out/r06_gputest_call55.txt; cat gpurun_out/r06_gputest_call55.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" | tail -2 | tee gpurun_out/r06_smoke_call55.txt
( time timeout 1500 python bench.py ) > gpurun_out/r06_bench_call55.json.log 2> gpurun_out/r06_bench_call55.err
grep -v "$F" gpurun_out/r06_bench_call55.err | tail -4
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r06_bench_call55.json.log').read().strip().splitlines()[-1])
print('value',j['value'],'ms/step',j['ms_per_step'],'ttft_p50',j.get('ttft_p50_ms'),'roofline',j['roofline']['frac'],j['roofline']['us_per_launch'],'traffic',j['roofline']['traffic'])
b=j.get('bs32',{}); print('bs32',b.get('value'),b.get('ttft_p50_ms'),b.get('roofline',{}).get('frac'),b.get('roofline',{}).get('us_per_launch'),b.get('roofline',{}).get('traffic'))
b=j.get('bs128',{}); print('bs128',b.get('value'),json.dumps(b.get('roofline'))[:300])
print('dac',json.dumps(j.get('dac',{}).get('bf16_bs32'))[:500])
print('cpu',j['cpu_baseline']['value'],j.get('gpu_over_cpu'))
PY
cd /tmp
rm -rf /tmp/pb; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pb -o p -- python $R/bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > /tmp/pb.log 2>&1
python $R/tools/prof_report.py $(find /tmp/pb -name '*.db' | head -1) 30 1734 > $R/gpurun_out/r06_bench_bs1_rocprof_summary_v6.txt 2>&1
cp $(find /tmp/pb -name '*stats*.csv' | head -3) $R/gpurun_out/ 2>/dev/null
rm -rf /tmp/pp; PROF_B=1 PROF_N=20 timeout 400 rocprofv3 --kernel-trace -d /tmp/pp -o p -- python $R/tools/prof_prefill.py > /tmp/pp.log 2>&1
python $R/tools/prof_report.py $(find /tmp/pp -name '*.db' | head -1) 30 > $R/gpurun_out/r06_prefill_kernels_bs1_v6.txt 2>&1
rm -rf /tmp/pp32; PROF_B=32 PROF_N=8 timeout 400 rocprofv3 --kernel-trace -d /tmp/pp32 -o p -- python $R/tools/prof_prefill.py > /tmp/pp32.log 2>&1
python $R/tools/prof_report.py $(find /tmp/pp32 -name '*.db' | head -1) 30 > $R/gpurun_out/r06_prefill_kernels_bs32_v9.txt 2>&1
for B in 32 128; do
  rm -rf /tmp/pl$B; PROF_B=$B PROF_STEPS=20 timeout 400 rocprofv3 --kernel-trace -d /tmp/pl$B -o p -- python $R/tools/prof_step.py > /dev/null 2>&1
  python $R/tools/prof_report.py $(find /tmp/pl$B -name '*.db' | head -1) 14 420 > $R/gpurun_out/r06_step_bf16_bs${B}_v6.txt 2>&1
done
cd $R
head -14 gpurun_out/r06_bench_bs1_rocprof_summary_v6.txt | cut -c1-150; tail -2 gpurun_out/r06_bench_bs1_rocprof_summary_v6.txt
head -16 gpurun_out/r06_step_bf16_bs32_v6.txt | cut -c1-150
