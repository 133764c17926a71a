#!/bin/bash
# round 5, GPU call 22: the prefill's K / V cache rows written by the fused LN1 + QKV node (no kv_append node): bit-identity + first-token time, LM tests on the path
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
{
timeout 120 tools/cabi_probe lm 1 tag=kv_in_qkv dump=/tmp/a.bin steps=4
PTTS_KV_IN_QKV=0 timeout 120 tools/cabi_probe lm 1 tag=kv_append_node dump=/tmp/b.bin steps=4
tools/cabi_probe cmp /tmp/a.bin /tmp/b.bin | tail -2
timeout 120 tools/cabi_probe lm 1 fp32 tag=kv_in_qkv dump=/tmp/a32.bin steps=4
PTTS_KV_IN_QKV=0 timeout 120 tools/cabi_probe lm 1 fp32 tag=kv_append_node dump=/tmp/b32.bin steps=4
tools/cabi_probe cmp /tmp/a32.bin /tmp/b32.bin | tail -2
} > gpurun_out/r05_probes22.txt 2>&1
grep -E "cabi_probe" gpurun_out/r05_probes22.txt | cut -c1-260
( timeout 600 python -m pytest tests/test_lm_gpu.py -m gpu -x -q -k "golden or voice_prompt or mini_width_two_layers or prefill or grouped_query" 2>&1 | grep -E "passed|failed|Error" | tail -3 ) > gpurun_out/r05_gputest22.txt
cat gpurun_out/r05_gputest22.txt
