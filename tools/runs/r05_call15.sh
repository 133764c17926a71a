#!/bin/bash
# round 5, GPU call 15: the last transposed conv (192 -> 96) on the LDS-tiled kernel with 32-channel staging chunks (228 VGPRs, two waves per SIMD) instead of the
# 96-channel instance (324 VGPRs, one wave per SIMD); k1 convs on the LDS-tiled kernel by default; stage parity with the switches on
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for B in 32 1; do
  timeout 120 tools/cabi_probe dac $B tag=default_k1_lds
  PTTS_DAC_NO_LDS_K1=1 timeout 120 tools/cabi_probe dac $B tag=k1_direct
  PTTS_DAC_LAST_UP_LDS=1 timeout 120 tools/cabi_probe dac $B tag=last_up_lds_ks1
  PTTS_DAC_LAST_UP_LDS=1 PTTS_DAC_UP2_KS3=1 timeout 120 tools/cabi_probe dac $B tag=last_up_lds_ks3
done
} > gpurun_out/r05_probes15.txt 2>&1
cat gpurun_out/r05_probes15.txt | cut -c1-200
( PTTS_DAC_LAST_UP_LDS=1 timeout 600 python -m pytest tests/test_dac_stage_parity_gpu.py -m gpu -x -q 2>&1 | tail -4 ) > gpurun_out/r05_gputest15.txt
cat gpurun_out/r05_gputest15.txt
