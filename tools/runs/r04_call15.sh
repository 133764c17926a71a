#!/bin/bash
# round 4, GPU call 15: single-utterance fused node (qkv_attn_kernel: LN1 + QKV rows + self-attention + append): parity, then step time A/B
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_lm_gpu.py -m gpu -q -x -k "fused_qkv_attention or single_utterance or layernorm_plus_projection_as_one_node" 2>&1 | tail -12 ) > gpurun_out/r04_gputest15.txt
{
timeout 120 tools/cabi_probe lm 1 tag=fuse_qa
PTTS_NO_FUSE_QA=1 timeout 120 tools/cabi_probe lm 1 tag=two_nodes
timeout 120 tools/cabi_probe lm 1 tag=fuse_qa
PTTS_NO_FUSE_QA=1 timeout 120 tools/cabi_probe lm 1 tag=two_nodes
PTTS_ATTN_SPLITS=2 timeout 120 tools/cabi_probe lm 1 tag=fuse_qa_s2
PTTS_ATTN_SPLITS=8 timeout 120 tools/cabi_probe lm 1 tag=fuse_qa_s8
timeout 120 tools/cabi_probe lm 1 large tag=fuse_qa
PTTS_NO_FUSE_QA=1 timeout 120 tools/cabi_probe lm 1 large tag=two_nodes
timeout 120 tools/cabi_probe lm 1 large fp8 tag=fuse_qa
PTTS_NO_FUSE_QA=1 timeout 120 tools/cabi_probe lm 1 large fp8 tag=two_nodes
timeout 120 tools/cabi_probe lm 32 tag=lnproj_default
} > gpurun_out/r04_probes15.txt 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r04_gputest15.txt | head; cat gpurun_out/r04_probes15.txt | cut -c1-120
