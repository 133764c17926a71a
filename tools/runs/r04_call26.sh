#!/bin/bash
# round 4, GPU call 26: with the lighter M passes, is the two-sub-batch split at 64..127 utterances still a win? + batch parity suite + bs=32 / 64 / 128 e2e
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_bench_config_parity_gpu.py tests/test_lm_gpu.py tests/test_generate_gpu.py -m gpu -q -x -k "decode_batch_above_32 or batch_above_8 or bf16_logits_and_argmax or layernorm_plus_projection or stream_split or free_running_graph_path_batch_12 or fp8 or e4m3" 2>&1 | tail -4 ) > gpurun_out/r04_gputest26.txt
{
STREAMS_PROBE_MAX_N=2 timeout 600 python tools/streams_probe.py 64 96 128
for B in 12 16 20 40 56 60; do timeout 120 tools/cabi_probe lm $B tag=policy; done
} > gpurun_out/r04_probes26.txt 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r04_gputest26.txt | head; grep -v "^$\|amdgpu.ids" gpurun_out/r04_probes26.txt | cut -c1-130
