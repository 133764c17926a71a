#!/bin/bash
# round 5, GPU call 18: 64-frame tiles (FT = 4 instances of conv_lds_kernel) where a launch has fewer than two workgroups per CU: single-utterance decode,
# a streamer-sized window, batch 32 (unchanged path); DAC parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for T in 860 56 200; do
  timeout 120 tools/cabi_probe dac 1 frames=$T reps=20 tag=ft4
  PTTS_DAC_NO_FT4=1 timeout 120 tools/cabi_probe dac 1 frames=$T reps=20 tag=ft8_only
done
timeout 120 tools/cabi_probe dac 2 frames=860 tag=ft4
PTTS_DAC_NO_FT4=1 timeout 120 tools/cabi_probe dac 2 frames=860 tag=ft8_only
timeout 120 tools/cabi_probe dac 32 tag=ft4
} > gpurun_out/r05_probes18.txt 2>&1
cat gpurun_out/r05_probes18.txt | cut -c1-200
( timeout 900 python -m pytest tests/test_dac_stage_parity_gpu.py tests/test_dac_gpu.py -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/r05_gputest18.txt
cat gpurun_out/r05_gputest18.txt
