#!/bin/bash
# round 6, GPU call 38: prefill_attn_mfma_kernel / t5_attn_mfma_kernel with every global load unconditional on a clamped address (found with
# tools/isa_load_chains.py: per-lane conditional loads had been compiled into 26 / 40 branch + load + s_waitcnt vmcnt(0) groups in a row): A/B against the
# previous kernels (tools/variants/prevattn, built from HEAD) on one box, then the T5 / LM / generate / bench-config suites
cd "$GRAFT_REPO_ROOT" || exit 1
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
N=tools/cabi_probe; O=tools/variants/prevattn/cabi_probe_prevattn
{
for i in 1 2 3; do
  for B in 32 64; do
    timeout 300 $N t5 $B tag=unconditional_loads 2>&1 | grep -v "$F" | cut -c1-110
    timeout 300 $O t5 $B tag=previous 2>&1 | grep -v "$F" | cut -c1-110
  done
  for cfg in "32" "64" "32 large"; do
    timeout 300 $N lm $cfg tag=unconditional_loads 2>&1 | grep -v "$F" | sed 's/us\/step.*prefill+first/... prefill+first/' | cut -c1-160
    timeout 300 $O lm $cfg tag=previous 2>&1 | grep -v "$F" | sed 's/us\/step.*prefill+first/... prefill+first/' | cut -c1-160
  done
done
} | tee gpurun_out/r06_attn_loads_ab.txt
timeout 2400 python -m pytest tests/test_t5_gpu.py tests/test_lm_gpu.py tests/test_generate_gpu.py tests/test_bench_config_parity_gpu.py -q -m gpu 2>&1 | grep -v "$F" | tail -6 | tee gpurun_out/r06_gputest_call38.txt
