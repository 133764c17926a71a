#!/bin/bash
# round 4, GPU call 14: LayerNorm + projection as one node (lnproj_fused_kernel, PTTS_LNPROJ = 0..3, G = 8 / 4) at batch 32 / 64 / 128: parity, then step time
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_lm_gpu.py -m gpu -q -x -k "layernorm_plus_projection" 2>&1 | tail -12 ) > gpurun_out/r04_gputest14.txt
{
for B in 32 64 128; do
  timeout 120 tools/cabi_probe lm $B tag=lnproj0
  PTTS_LNPROJ=1 timeout 120 tools/cabi_probe lm $B tag=lnproj1_g8
  PTTS_LNPROJ=2 timeout 120 tools/cabi_probe lm $B tag=lnproj2_g8
  PTTS_LNPROJ=3 timeout 120 tools/cabi_probe lm $B tag=lnproj3_g8
  PTTS_LNPROJ=3 PTTS_LNPROJ_G=4 timeout 120 tools/cabi_probe lm $B tag=lnproj3_g4
done
PTTS_LNPROJ=3 timeout 120 tools/cabi_probe lm 12 tag=lnproj3_g8
timeout 120 tools/cabi_probe lm 12 tag=lnproj0
} > gpurun_out/r04_probes14.txt 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r04_gputest14.txt | head; cat gpurun_out/r04_probes14.txt | cut -c1-110
