#!/bin/bash
# round 5, GPU call 9: kernarg preload probe (tools/kernarg_probe.hip): dependent graph nodes with by-value struct arguments (s_load) vs scalar
# arguments preloaded into user SGPRs by the CP (-mllvm -amdgpu-kernarg-preload-count=14)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
timeout 120 tools/kernarg_probe_base
timeout 120 tools/kernarg_probe_pre
timeout 120 tools/kernarg_probe_base
timeout 120 tools/kernarg_probe_pre
} > gpurun_out/r05_probes9.txt 2>&1
cat gpurun_out/r05_probes9.txt
