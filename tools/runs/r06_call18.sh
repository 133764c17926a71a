#!/bin/bash
# round 6, GPU call 18: VERDICT r05 item 3a - phase stamps of the seven MFMA-path nodes per layer of the batch > 8 step (measurement build)
cd "$GRAFT_REPO_ROOT" || exit 1
for B in 32 128; do
  timeout 300 tools/stamps/cabi_probe_stamps lm $B tag=stamps 2>&1 | grep -v "amdgpu.ids"
done | tee gpurun_out/r06_node_stamps_raw_v2.txt
