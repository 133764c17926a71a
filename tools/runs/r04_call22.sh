#!/bin/bash
# round 4, GPU call 22: xq_attn_kernel (un-folded cross block's LN2 + q rows + cross-attention as one node, 1..8 utterances on the GEMV step): parity, step time A/B
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_lm_gpu.py tests/test_generate_gpu.py -m gpu -q -x -k "fused_cross_q or gemv_step or batch_2_to_8 or fp8 or e4m3 or generate" 2>&1 | tail -6 ) > gpurun_out/r04_gputest22.txt
{
for B in 2 4 8; do
  timeout 120 tools/cabi_probe lm $B tag=xq
  PTTS_NO_FUSE_XQ=1 timeout 120 tools/cabi_probe lm $B tag=two_nodes
done
timeout 120 tools/cabi_probe lm 4 large fp8 tag=xq
PTTS_NO_FUSE_XQ=1 timeout 120 tools/cabi_probe lm 4 large fp8 tag=two_nodes
timeout 120 tools/cabi_probe lm 8 large tag=xq
PTTS_NO_FUSE_XQ=1 timeout 120 tools/cabi_probe lm 8 large tag=two_nodes
PTTS_NO_XFOLD=1 timeout 120 tools/cabi_probe lm 1 tag=nofold_xq
PTTS_NO_XFOLD=1 PTTS_NO_FUSE_XQ=1 timeout 120 tools/cabi_probe lm 1 tag=nofold_two_nodes
} > gpurun_out/r04_probes22.txt 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r04_gputest22.txt | head; cat gpurun_out/r04_probes22.txt | cut -c1-120
