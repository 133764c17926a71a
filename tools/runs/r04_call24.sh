#!/bin/bash
# round 4, GPU call 24: strip GEMMs above 32 utterances in 32-row instead of 64-row M passes (PTTS_MSPLIT_ROWS=32): parity, A/B at 64 / 128
export TMPDIR=/tmp
mkdir -p gpurun_out
( PTTS_MSPLIT_ROWS=32 timeout 900 python -m pytest tests/test_bench_config_parity_gpu.py -m gpu -q -x -k "decode_batch_above_32" 2>&1 | tail -4 ) > gpurun_out/r04_gputest24.txt
{
for B in 64 128; do
  timeout 120 tools/cabi_probe lm $B tag=rows64
  PTTS_MSPLIT_ROWS=32 timeout 120 tools/cabi_probe lm $B tag=rows32
  timeout 120 tools/cabi_probe lm $B tag=rows64
  PTTS_MSPLIT_ROWS=32 timeout 120 tools/cabi_probe lm $B tag=rows32
done
} > gpurun_out/r04_probes24.txt 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r04_gputest24.txt | head; cat gpurun_out/r04_probes24.txt | cut -c1-120
