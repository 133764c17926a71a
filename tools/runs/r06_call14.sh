#!/bin/bash
# round 6, GPU call 14: attn_kernel's first K / V batch addressed from kernel arguments only (default) vs clamped by the device-resident length (PTTS_ATTN_EXACT=1)
cd "$GRAFT_REPO_ROOT" || exit 1
for B in 32 128 16; do
  timeout 300 tools/cabi_probe lm $B tag=first_batch_from_kernargs dump=/tmp/a$B.bin 2>&1 | grep -v "amdgpu.ids"
  PTTS_ATTN_EXACT=1 timeout 300 tools/cabi_probe lm $B tag=first_batch_exact_len dump=/tmp/b$B.bin 2>&1 | grep -v "amdgpu.ids"
  tools/cabi_probe cmp /tmp/a$B.bin /tmp/b$B.bin 2>&1 | tail -2
done | tee gpurun_out/r06_attn_first_batch.txt
