#!/bin/bash
# round 6, GPU call 16: fc2 split-K 4 (product) vs un-split on the other engines that split today: e4m3 weights, Large-v1 widths, 12 / 24 utterances
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "12" "24" "16 fp8" "32 fp8" "16 large" "32 large" "32 large fp8"; do
  timeout 300 tools/cabi_probe lm $cfg tag=ksplit4 2>&1 | grep -v "amdgpu.ids" | cut -c1-120
  timeout 300 tools/variants/ks0/cabi_probe_ks0 lm $cfg tag=unsplit 2>&1 | grep -v "amdgpu.ids" | cut -c1-120
done | tee gpurun_out/r06_fc2_ksplit_2.txt
