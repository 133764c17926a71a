#!/bin/bash
# round 4, GPU call 7: conv_lds epilogue through LDS + last transposed conv on conv_lds (A/B), DAC tests, kernel table
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for B in 1 32; do
  timeout 120 tools/cabi_probe dac $B tag=default
  PTTS_DAC_CONV_EPI_DIRECT=1 timeout 120 tools/cabi_probe dac $B tag=conv_epi_direct
  PTTS_DAC_LAST_UP_DIRECT=1 timeout 120 tools/cabi_probe dac $B tag=last_up_direct
  PTTS_DAC_LAST_UP_DIRECT=1 PTTS_DAC_CONV_EPI_DIRECT=1 timeout 120 tools/cabi_probe dac $B tag=both_direct
done
} > gpurun_out/r04_probes7.txt 2>&1
cd /tmp
rm -rf /tmp/pd32; timeout 300 rocprofv3 --kernel-trace -d /tmp/pd32 -o p -- $GRAFT_REPO_ROOT/tools/cabi_probe dac 32 reps=3 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/prof_report.py $(find /tmp/pd32 -name '*.db' | head -1) 14 > $GRAFT_REPO_ROOT/gpurun_out/r04_dac_kernels_bs32_v5.txt 2>&1
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_dac_stage_parity_gpu.py tests/test_dac_gpu.py tests/test_streamer_gpu.py -m gpu -q 2>&1 | tail -15 ) > gpurun_out/r04_gputest7.txt
tail -5 gpurun_out/r04_gputest7.txt; cat gpurun_out/r04_probes7.txt | cut -c1-160; head -14 gpurun_out/r04_dac_kernels_bs32_v5.txt
