#!/bin/bash
# round 4, GPU call 16: single-utterance fused cross node (xfold_attn_kernel + GV_LNP): parity, then step time A/B
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_lm_gpu.py tests/test_generate_gpu.py -m gpu -q -x -k "fused_cross or fused_qkv or single_utterance or fp8 or e4m3 or generate" 2>&1 | tail -12 ) > gpurun_out/r04_gputest16.txt
{
timeout 120 tools/cabi_probe lm 1 tag=fuse_x_nur2
PTTS_NO_FUSE_X=1 timeout 120 tools/cabi_probe lm 1 tag=x_two_nodes
PTTS_FUSE_X_NUR=4 timeout 120 tools/cabi_probe lm 1 tag=fuse_x_nur4
timeout 120 tools/cabi_probe lm 1 tag=fuse_x_nur2
PTTS_NO_FUSE_X=1 timeout 120 tools/cabi_probe lm 1 tag=x_two_nodes
timeout 120 tools/cabi_probe lm 1 large tag=fuse_x_nur2
PTTS_FUSE_X_NUR=4 timeout 120 tools/cabi_probe lm 1 large tag=fuse_x_nur4
PTTS_NO_FUSE_X=1 timeout 120 tools/cabi_probe lm 1 large tag=x_two_nodes
timeout 120 tools/cabi_probe lm 1 large fp8 tag=fuse_x_nur2
PTTS_NO_FUSE_X=1 timeout 120 tools/cabi_probe lm 1 large fp8 tag=x_two_nodes
} > gpurun_out/r04_probes16.txt 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r04_gputest16.txt | head; cat gpurun_out/r04_probes16.txt | cut -c1-120
