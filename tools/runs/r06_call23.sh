#!/bin/bash
# round 6, GPU call 23 (re-entry): gemm_strip_kernel preloaded scalars vs by-value struct - correctness on the small golden specs + step times, one box
cd "$GRAFT_REPO_ROOT" || exit 1
{
echo "---- product (preloaded scalars)"; python tools/small_spec_probe.py 2>&1 | grep -v amdgpu.ids
echo "---- by-value variant"; PTTS_LIB=$PWD/tools/variants/byval/libptts_hip_byval.so python tools/small_spec_probe.py 2>&1 | grep -v amdgpu.ids | tail -12
for cfg in "16" "32" "64" "128" "32 fp8" "32 large"; do
  timeout 300 tools/cabi_probe lm $cfg tag=preload 2>&1 | grep -v "amdgpu.ids" | cut -c1-120
  timeout 300 tools/variants/byval/cabi_probe_byval lm $cfg tag=byvalue 2>&1 | grep -v "amdgpu.ids" | cut -c1-120
done
} | tee gpurun_out/r06_strip_preload_ab.txt
