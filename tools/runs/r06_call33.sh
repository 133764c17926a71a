#!/bin/bash
# round 6, GPU call 33: one-workgroup-per-CU tiles for T5's q|k|v (192 x 128) and wi (176 x 256) projections at 32 descriptions: A/B through the dev-knob
# build (PTTS_GLDS_BIG_TILES=0 = the calls 2-6 policy), then the whole GPU suite on the product library (XIN residual units by default, new tiles)
cd "$GRAFT_REPO_ROOT" || exit 1
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
P=tools/variants/t5dev/cabi_probe_t5dev
{
for i in 1 2 3; do
  for B in 32 16 64; do
    timeout 300 $P t5 $B tag=big_tiles 2>&1 | grep -v "$F" | cut -c1-200
    PTTS_GLDS_BIG_TILES=0 timeout 300 $P t5 $B tag=calls2-6 2>&1 | grep -v "$F" | cut -c1-200
  done
done
} | tee gpurun_out/r06_t5_big_tiles_ab.txt
timeout 2700 python -m pytest tests -q -m gpu 2>&1 | grep -v "$F" | tail -12 | tee gpurun_out/r06_gputest_call33.txt
