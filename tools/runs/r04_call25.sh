#!/bin/bash
# round 4, GPU call 25: rows per M pass of the strip GEMMs (16 / 32 / 64) at 24 / 32 / 48 / 64 / 96 / 128 utterances; parity with 16-row and 32-row passes
export TMPDIR=/tmp
mkdir -p gpurun_out
( PTTS_MSPLIT_ROWS=16 timeout 900 python -m pytest tests/test_bench_config_parity_gpu.py tests/test_lm_gpu.py -m gpu -q -x -k "decode_batch_above_32 or batch_above_8 or bf16_logits_and_argmax or layernorm_plus_projection" 2>&1 | tail -4 ) > gpurun_out/r04_gputest25.txt
{
for B in 24 32 48 64 96 128; do
  for R in 64 32 16; do PTTS_MSPLIT_ROWS=$R timeout 120 tools/cabi_probe lm $B tag=rows$R; done
done
for R in 64 32 16; do PTTS_MSPLIT_ROWS=$R timeout 120 tools/cabi_probe lm 32 large fp8 tag=rows$R; done
} > gpurun_out/r04_probes25.txt 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r04_gputest25.txt | head; cat gpurun_out/r04_probes25.txt | cut -c1-110
