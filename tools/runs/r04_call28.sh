#!/bin/bash
# round 4, GPU call 28: prefill rows - lighter M passes only where the projection is short of workgroups (PTTS_MSPLIT_PREFILL=2) vs everywhere (1) vs never (0)
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for A in "1" "2" "4" "1 large" "1 large fp8" "1 ctx=100" "1 fp32"; do
  for V in 0 1 2 0 1 2; do PTTS_MSPLIT_PREFILL=$V timeout 120 tools/cabi_probe lm $A tag=x | sed -e 's/.*prefill+first token/prefill+first token/' -e 's/; create.*//' -e "s/^/[$A] policy=$V: /"; done
done
} > gpurun_out/r04_probes28.txt 2>&1
PTTS_MSPLIT_PREFILL=2 timeout 600 python -m pytest tests/test_lm_gpu.py -m gpu -q -x -k "prefill or voice or golden" 2>&1 | tail -2 >> gpurun_out/r04_probes28.txt
cat gpurun_out/r04_probes28.txt | cut -c1-120
