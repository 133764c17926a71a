#!/bin/bash
# round 4, GPU call 11: DAC decode time per utterance vs sub-batch size (do smaller sub-batches keep the inter-layer activations in the Infinity Cache?)
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for B in 1 2 3 4 6 8 12 16 24 32; do timeout 120 tools/cabi_probe dac $B tag=subbatch; done
} > gpurun_out/r04_probes11.txt 2>&1
( timeout 600 python -m pytest tests/test_generate_gpu.py tests/test_dac_gpu.py -m gpu -q -k "stream_split or fused or ragged" 2>&1 | grep -E "passed|failed" ) >> gpurun_out/r04_probes11.txt
cat gpurun_out/r04_probes11.txt | cut -c1-140
