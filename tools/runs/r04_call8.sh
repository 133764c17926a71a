#!/bin/bash
# round 4, GPU call 8: timing ablations of the fused residual units (PTTS_DAC_DBG: results wrong on purpose), final DAC numbers, DAC tests on the final kernels
export TMPDIR=/tmp
mkdir -p gpurun_out
{
# (needs the library of commit 46a93c0: the PTTS_DAC_DBG ablation switches were removed afterwards)
for D in 0 1 2 4 8 5 15; do PTTS_DAC_DBG=$D timeout 120 tools/cabi_probe dac 32 tag=dbg$D; done
timeout 120 tools/cabi_probe dac 1 tag=final
} > gpurun_out/r04_probes8.txt 2>&1
( timeout 900 python -m pytest tests/test_dac_stage_parity_gpu.py tests/test_dac_gpu.py -m gpu -q 2>&1 | tail -8 ) > gpurun_out/r04_gputest8.txt
tail -4 gpurun_out/r04_gputest8.txt; cat gpurun_out/r04_probes8.txt | cut -c1-150
