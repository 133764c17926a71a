#!/bin/bash
# round 4, GPU call 23: final LayerNorm + LM heads as one lnproj node at 9..40 utterances: parity, A/B
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_lm_gpu.py tests/test_bench_config_parity_gpu.py -m gpu -q -x -k "layernorm_plus_projection or decode_batch_above_32 or batch_above_8 or bf16_logits_and_argmax or free_running_graph_path_batch_12 or several_steps" 2>&1 | tail -6 ) > gpurun_out/r04_gputest23.txt
{
for B in 12 32; do
  timeout 120 tools/cabi_probe lm $B tag=lnproj_heads
  PTTS_NO_LNPROJ_HEADS=1 timeout 120 tools/cabi_probe lm $B tag=prep_plus_gemm
  timeout 120 tools/cabi_probe lm $B tag=lnproj_heads
  PTTS_NO_LNPROJ_HEADS=1 timeout 120 tools/cabi_probe lm $B tag=prep_plus_gemm
done
} > gpurun_out/r04_probes23.txt 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r04_gputest23.txt | head; cat gpurun_out/r04_probes23.txt | cut -c1-120
