#!/bin/bash
# round 6, GPU call 15: fc2 split-K factor at 9..40 utterances: 4 (product) vs 2 vs un-split (the stamps of call 13 show what the partials cost downstream)
cd "$GRAFT_REPO_ROOT" || exit 1
for B in 16 32 40; do
  timeout 300 tools/cabi_probe lm $B tag=ksplit4 2>&1 | grep -v "amdgpu.ids"
  timeout 300 tools/variants/ks2/cabi_probe_ks2 lm $B tag=ksplit2 2>&1 | grep -v "amdgpu.ids"
  timeout 300 tools/variants/ks0/cabi_probe_ks0 lm $B tag=unsplit 2>&1 | grep -v "amdgpu.ids"
done | tee gpurun_out/r06_fc2_ksplit.txt
