#!/bin/bash
# round 4, GPU call 9: Snake with the per-channel inverse precomputed; DAC tests (goldens: exact-f32 path must stay bit-compatible), timings
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for B in 1 32; do timeout 120 tools/cabi_probe dac $B tag=inv_alpha; done
PTTS_DAC_DBG=8 timeout 120 tools/cabi_probe dac 32 tag=dbg8_nosnake  # (library of that commit: ablation switch since removed)
timeout 120 tools/cabi_probe dac 1 f32 tag=f32
} > gpurun_out/r04_probes9.txt 2>&1
( timeout 900 python -m pytest tests/test_dac_stage_parity_gpu.py tests/test_dac_gpu.py tests/test_generate_gpu.py -m gpu -q 2>&1 | tail -8 ) > gpurun_out/r04_gputest9.txt
tail -4 gpurun_out/r04_gputest9.txt; cat gpurun_out/r04_probes9.txt | cut -c1-150
