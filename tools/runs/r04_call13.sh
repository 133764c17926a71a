#!/bin/bash
# round 4, GPU call 13: decode self-attention with 16 K/V row groups per wave in flight (PTTS_ATTN_U=16) at batch 32 / 64 / 128
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for B in 32 64 128; do
  timeout 120 tools/cabi_probe lm $B tag=u8
  PTTS_ATTN_U=16 timeout 120 tools/cabi_probe lm $B tag=u16
done
} > gpurun_out/r04_probes13.txt 2>&1
( timeout 600 python -m pytest tests/test_lm_gpu.py -m gpu -q -k "16_row_groups or long_context" 2>&1 | grep -E "passed|failed|Error" ) >> gpurun_out/r04_probes13.txt
cat gpurun_out/r04_probes13.txt | cut -c1-140
