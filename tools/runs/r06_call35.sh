#!/bin/bash
# round 6, GPU call 35: big-tile policy over whole rounds (T5 at 64 / 128 descriptions) A/B through the dev-knob build; gemm_probe on the prefill shapes with
# 48- / 96-row tiles (1056 rows = an even number of 48-row tiles: xcd_tile_order can split them 2 x 4); T5 + LM + generate + bench-config suites
cd "$GRAFT_REPO_ROOT" || exit 1
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
P=tools/variants/t5dev/cabi_probe_t5dev
{
for i in 1 2; do
  for B in 32 64 128; do
    timeout 300 $P t5 $B tag=big_tiles 2>&1 | grep -v "$F" | cut -c1-120
    PTTS_GLDS_BIG_TILES=0 timeout 300 $P t5 $B tag=calls2-6 2>&1 | grep -v "$F" | cut -c1-120
  done
done
} | tee gpurun_out/r06_t5_big_tiles_ab2.txt
timeout 900 tools/gemm_probe 40 "prefill" 2>&1 | grep -v "$F" | tee gpurun_out/r06_gemm_probe_call35.txt | tail -3
timeout 2400 python -m pytest tests/test_t5_gpu.py tests/test_lm_gpu.py tests/test_generate_gpu.py tests/test_bench_config_parity_gpu.py tests/test_dac_gpu.py -q -m gpu 2>&1 | grep -v "$F" | tail -6 | tee gpurun_out/r06_gputest_call35.txt
