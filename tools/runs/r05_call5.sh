#!/bin/bash
# round 5, GPU call 5: tiled prefill attention + LDS-tiled GEMM for prefill-sized rows (parity: whole suite; time to the first token at 1 / 32 utterances,
# A/B of each), node stamps of the single-utterance step on the device-wide clock (wall_clock64)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
{
timeout 120 tools/stamps/cabi_probe_stamps lm 1 tag=stamps
timeout 120 tools/stamps/cabi_probe_stamps lm 1 tag=stamps_again
} > gpurun_out/r05_node_stamps.txt 2>&1
cut -c1-330 gpurun_out/r05_node_stamps.txt
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest5.txt 2>&1; echo "suite rc=$?" >> gpurun_out/r05_gputest5.txt
tail -12 gpurun_out/r05_gputest5.txt
{
timeout 300 python tools/ttft_probe5.py tiled_attn+tile_gemm
PTTS_GEMM_TILE=0 timeout 300 python tools/ttft_probe5.py tiled_attn+block_gemm
PTTS_PREFILL_ATTN=0 PTTS_GEMM_TILE=0 timeout 300 python tools/ttft_probe5.py round4_prefill
for t in 88 48 84 44; do PTTS_GEMM_TILE=$t timeout 300 python tools/ttft_probe5.py tile$t 32; done
} > gpurun_out/r05_probes5.txt 2>&1
grep ttft_probe5 gpurun_out/r05_probes5.txt | cut -c1-330
