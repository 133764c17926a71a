#!/bin/bash
# round 4, GPU call 27: the lighter M passes on PREFILL rows too (time-to-first-token path)? tools/cabi_probe prints prefill + first token of the second call
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for B in 1 2 4 7; do
  for V in 0 1 0 1; do PTTS_MSPLIT_PREFILL=$V timeout 120 tools/cabi_probe lm $B tag=prefill_policy$V | sed -e 's/.*prefill+first token/prefill+first token/' -e "s/^/B=$B policy=$V: /"; done
done
for V in 0 1; do PTTS_MSPLIT_PREFILL=$V timeout 120 tools/cabi_probe lm 1 large tag=x | sed -e 's/.*prefill+first token/prefill+first token/' -e "s/^/large B=1 policy=$V: /"; done
for V in 0 1; do PTTS_MSPLIT_PREFILL=$V timeout 120 tools/cabi_probe lm 1 ctx=100 tag=x | sed -e 's/.*prefill+first token/prefill+first token/' -e "s/^/B=1 prompt 100 policy=$V: /"; done
} > gpurun_out/r04_probes27.txt 2>&1
PTTS_MSPLIT_PREFILL=1 timeout 600 python -m pytest tests/test_lm_gpu.py -m gpu -q -x -k "prefill or voice or golden" 2>&1 | tail -2 >> gpurun_out/r04_probes27.txt
cat gpurun_out/r04_probes27.txt | cut -c1-150
