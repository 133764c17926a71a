#!/bin/bash
# round 4, GPU call 17: qkv_attn_kernel as 8 waves (wave 0 normalises while its weights fly), U = 8 row groups in flight, fp32 at H = 1024 (parity engine)
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_lm_gpu.py -m gpu -q -x -k "fused_cross or fused_qkv or single_utterance" 2>&1 | tail -6 ) > gpurun_out/r04_gputest17.txt
( timeout 1200 python -m pytest tests/test_bench_config_parity_gpu.py -m gpu -q -x -k "fp32_bs1_all_868 or bf16_logits_and_argmax" 2>&1 | tail -6 ) >> gpurun_out/r04_gputest17.txt
{
timeout 120 tools/cabi_probe lm 1 tag=u8
PTTS_FUSE_QA_U=4 timeout 120 tools/cabi_probe lm 1 tag=u4
timeout 120 tools/cabi_probe lm 1 tag=u8
PTTS_FUSE_QA_U=4 timeout 120 tools/cabi_probe lm 1 tag=u4
PTTS_NO_FUSE_QA=1 timeout 120 tools/cabi_probe lm 1 tag=two_nodes
timeout 120 tools/cabi_probe lm 1 fp32 tag=fused_u4
PTTS_NO_FUSE_QA=1 timeout 120 tools/cabi_probe lm 1 fp32 tag=two_nodes
timeout 120 tools/cabi_probe lm 1 large tag=u8
PTTS_FUSE_QA_U=4 timeout 120 tools/cabi_probe lm 1 large tag=u4
timeout 120 tools/cabi_probe lm 1 large fp8 tag=u8
PTTS_FUSE_QA_U=4 timeout 120 tools/cabi_probe lm 1 large fp8 tag=u4
} > gpurun_out/r04_probes17.txt 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r04_gputest17.txt | head; cat gpurun_out/r04_probes17.txt | cut -c1-120
