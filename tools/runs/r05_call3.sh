#!/bin/bash
# round 5, GPU call 3: whole GPU suite on the new defaults (16-row LayerNorm + projection nodes above 40 utterances, lnproj on the prefill rows, fused
# self-attention node up to 3 utterances, e4m3 KV cache mode), then: e4m3 KV cache step times at 32 / 64 / 128 utterances, rows per M pass with the new nodes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest3.txt 2>&1; echo "suite rc=$?" >> gpurun_out/r05_gputest3.txt
tail -12 gpurun_out/r05_gputest3.txt
{
for B in 32 64 128; do
  timeout 120 tools/cabi_probe lm $B tag=bf16_cache
  timeout 120 tools/cabi_probe lm $B kv8 tag=e4m3_cache
done
for B in 64 96 128; do
  for R in 16 32 64; do PTTS_MSPLIT_ROWS=$R timeout 120 tools/cabi_probe lm $B tag=rows$R; done
done
timeout 120 tools/cabi_probe lm 32 large tag=default
timeout 120 tools/cabi_probe lm 64 large tag=default
PTTS_LNPROJ_G=8 PTTS_LNPROJ=0 timeout 120 tools/cabi_probe lm 64 large tag=round4_nodes
} > gpurun_out/r05_probes3.txt 2>&1
cut -c1-120 gpurun_out/r05_probes3.txt
