#!/bin/bash
# round 4, GPU call 20: small batches as independent single-utterance engines on their own streams vs the batched GEMV step; xfold_attn_kernel with 8 waves
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_lm_gpu.py -m gpu -q -x -k "fused_cross" 2>&1 | tail -3 ) > gpurun_out/r04_gputest20.txt
{
timeout 120 tools/cabi_probe lm 1 tag=xfold8w
PTTS_FUSE_X=0 timeout 120 tools/cabi_probe lm 1 tag=x_two_nodes
timeout 120 tools/cabi_probe lm 1 tag=xfold8w
timeout 400 python tools/streams_probe_small.py mini 2 4 8
timeout 400 python tools/streams_probe_small.py large_fp8 4
} > gpurun_out/r04_probes20.txt 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r04_gputest20.txt | head; grep -v "^$" gpurun_out/r04_probes20.txt | cut -c1-150 | tail -25
