#!/bin/bash
# round 6, GPU call 30: the prefill's QKV projection above 256 rows writes the K / V cache rows in its tile epilogue (no kv_append node): A/B through the
# dev-knob build (PTTS_KV_IN_QKV=0 = the separate node) + parity suites
cd "$GRAFT_REPO_ROOT" || exit 1
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
{
for i in 1 2; do
for cfg in "32" "16" "64" "32 large"; do
  timeout 300 tools/variants/devk/cabi_probe_devk lm $cfg tag=kv_in_gemm 2>&1 | grep -v "$F" | sed 's/us\/step.*prefill+first/... prefill+first/' | cut -c1-160
  PTTS_KV_IN_QKV=0 timeout 300 tools/variants/devk/cabi_probe_devk lm $cfg tag=kv_append_node 2>&1 | grep -v "$F" | sed 's/us\/step.*prefill+first/... prefill+first/' | cut -c1-160
done
done
} | tee gpurun_out/r06_kv_in_gemm_ab.txt
timeout 2400 python -m pytest tests/test_lm_gpu.py tests/test_generate_gpu.py tests/test_bench_config_parity_gpu.py -q -m gpu 2>&1 | grep -v "$F" | tail -8 | tee gpurun_out/r06_gputest_call30.txt
