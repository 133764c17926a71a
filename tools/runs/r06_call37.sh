#!/bin/bash
# round 6, GPU call 37: the all-layers cross K / V launch on 128 x 128 tiles (the tile policy counted one layer's tiles: 256 -> 128 x 64; all 24 layers' tiles make
# it a 6144-tile launch): prefill + first token A/B through the dev-knob build (PTTS_GLDS_KV_TILES=0 = per-layer count)
cd "$GRAFT_REPO_ROOT" || exit 1
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
P=tools/variants/t5dev/cabi_probe_t5dev
{
for i in 1 2 3; do
for cfg in "32" "64" "32 large"; do
  timeout 300 $P lm $cfg tag=kv_all_layers 2>&1 | grep -v "$F" | sed 's/us\/step.*prefill+first/... prefill+first/' | cut -c1-160
  PTTS_GLDS_KV_TILES=0 timeout 300 $P lm $cfg tag=kv_per_layer 2>&1 | grep -v "$F" | sed 's/us\/step.*prefill+first/... prefill+first/' | cut -c1-160
done
done
} | tee gpurun_out/r06_kv_tiles_ab.txt
timeout 1200 python -m pytest tests/test_lm_gpu.py -q -m gpu -k "prefill or cross" 2>&1 | grep -v "$F" | tail -4 | tee gpurun_out/r06_gputest_call37.txt
