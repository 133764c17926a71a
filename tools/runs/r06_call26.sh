#!/bin/bash
# round 6, GPU call 26: lnproj_fused_kernel + xattn_fused_kernel on preloaded scalar arguments vs the committed tree (one box) + LM tests;
# per-kernel HBM traffic of the 32-utterance step and of the codec at 32 utterances (item 5: are the C = 96 units bandwidth-bound?)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
{
for cfg in "16" "32" "64" "128" "32 large" "1 ctx=32"; do
  timeout 300 tools/cabi_probe lm $cfg tag=lnproj+xattn_preload 2>&1 | grep -v "amdgpu.ids" | cut -c1-200
  timeout 300 tools/variants/base/cabi_probe_base lm $cfg tag=committed 2>&1 | grep -v "amdgpu.ids" | cut -c1-200
done
} | tee gpurun_out/r06_lnproj_xattn_preload_ab.txt
timeout 1500 python -m pytest tests/test_lm_gpu.py tests/test_generate_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee gpurun_out/r06_gputest_call26.txt
# per-kernel PMC, 32 utterances at the timed context
( cd /tmp && for c in FETCH_SIZE WRITE_SIZE; do PROF_B=32 PROF_STEPS=24 PROF_P=430 timeout 300 rocprofv3 --pmc $c -d /tmp/pmc32_$c -o p -- python $GRAFT_REPO_ROOT/tools/prof_eager.py > /dev/null 2>&1; done )
python tools/pmc_report2.py $(find /tmp/pmc32_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/pmc32_WRITE_SIZE -name "*.db" | head -1) 16 447 32 /tmp/t32.json 2>&1 | tee gpurun_out/r06_pmc_step_bs32.txt
# codec at 32 utterances: durations + FETCH / WRITE per kernel
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/dac_trace -o p -- $GRAFT_REPO_ROOT/tools/cabi_probe dac 32 reps=2 > /dev/null 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do timeout 300 rocprofv3 --pmc $c -d /tmp/dac_$c -o p -- $GRAFT_REPO_ROOT/tools/cabi_probe dac 32 reps=2 > /tmp/dac_$c.log 2>&1; done )
tools/cabi_probe dac 32 reps=2 2>&1 | grep -v amdgpu.ids | cut -c1-200
python tools/pmc_dac_report.py $(find /tmp/dac_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/dac_WRITE_SIZE -name "*.db" | head -1) $(find /tmp/dac_trace -name "*.db" | head -1) 3 2>&1 | tee gpurun_out/r06_pmc_dac_bs32.txt
