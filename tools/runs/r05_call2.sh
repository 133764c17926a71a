#!/bin/bash
# round 5, GPU call 2: fused self-attention node at 2..8 utterances (parity + step times), LayerNorm + projection node with 16 rows per workgroup at
# 48..128 utterances (A/B + dump compare), LayerNorm + projection nodes on the prefill rows (time to the first token)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_lm_gpu.py -m gpu -q -x -k "fused_qkv_attention_node or gemv or batch" 2>&1 | tail -6 ) > gpurun_out/r05_gputest2.txt
{
for B in 2 3 4 8; do
  timeout 120 tools/cabi_probe lm $B tag=fuse_multi
  PTTS_FUSE_QA_MULTI=0 timeout 120 tools/cabi_probe lm $B tag=two_nodes
done
timeout 120 tools/cabi_probe lm 8 large tag=fuse_multi
PTTS_FUSE_QA_MULTI=0 timeout 120 tools/cabi_probe lm 8 large tag=two_nodes
timeout 120 tools/cabi_probe lm 4 large fp8 tag=fuse_multi
PTTS_FUSE_QA_MULTI=0 timeout 120 tools/cabi_probe lm 4 large fp8 tag=two_nodes
for B in 48 64 96 128; do
  timeout 120 tools/cabi_probe lm $B tag=default dump=/tmp/d_$B.bin steps=4
  PTTS_LNPROJ=3 PTTS_LNPROJ_G=16 timeout 120 tools/cabi_probe lm $B tag=lnproj_g16 dump=/tmp/g_$B.bin steps=4
  PTTS_LNPROJ=1 PTTS_LNPROJ_G=16 timeout 120 tools/cabi_probe lm $B tag=lnproj_g16_ln1_only
  tools/cabi_probe cmp /tmp/d_$B.bin /tmp/g_$B.bin | tail -3
done
PTTS_LNPROJ=3 PTTS_LNPROJ_G=16 timeout 120 tools/cabi_probe lm 32 tag=lnproj_g16
timeout 120 tools/cabi_probe lm 32 tag=default
PTTS_LNPROJ_PREFILL=1 timeout 300 python tools/ttft_probe5.py lnproj_prefill 1
timeout 300 python tools/ttft_probe5.py default 1
} > gpurun_out/r05_probes2.txt 2>&1
cat gpurun_out/r05_gputest2.txt; cut -c1-400 gpurun_out/r05_probes2.txt
