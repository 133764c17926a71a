#!/bin/bash
# round 6, GPU call 29: the non-FULL strip instances on prepared rows (T5's wo projection, K = 2816) on the preloaded entry point: T5 encode time A/B + tests
cd "$GRAFT_REPO_ROOT" || exit 1
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
{
for i in 1 2; do
for cfg in "1" "4" "32"; do
  timeout 300 tools/cabi_probe t5 $cfg tag=wo_preloaded 2>&1 | grep -v "$F" | cut -c1-180
  timeout 300 tools/variants/prevt5/cabi_probe_prevt5 t5 $cfg tag=wo_by_value 2>&1 | grep -v "$F" | cut -c1-180
done
done
} | tee gpurun_out/r06_t5_wo_preload_ab.txt
timeout 2000 python -m pytest tests/test_t5_gpu.py tests/test_lm_gpu.py tests/test_generate_gpu.py -q -m gpu 2>&1 | grep -v "$F" | tail -8 | tee gpurun_out/r06_gputest_call29.txt
