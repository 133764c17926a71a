#!/bin/bash
# round 4, GPU call 4: deep weight prefetch in the fused residual units (A/B), the batch-1 persistent-layer probe, DAC tests
export TMPDIR=/tmp
mkdir -p gpurun_out
{
timeout 300 tools/persist_probe layers=24 reps=20
timeout 300 tools/persist_probe layers=8 reps=20
for B in 1 32; do
  timeout 120 tools/cabi_probe dac $B tag=wd3
  PTTS_DAC_WD1=1 timeout 120 tools/cabi_probe dac $B tag=wd1
done
} > gpurun_out/r04_probes4.txt 2>&1
cd /tmp
rm -rf /tmp/pd32; timeout 300 rocprofv3 --kernel-trace -d /tmp/pd32 -o p -- $GRAFT_REPO_ROOT/tools/cabi_probe dac 32 reps=3 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/prof_report.py $(find /tmp/pd32 -name '*.db' | head -1) 14 > $GRAFT_REPO_ROOT/gpurun_out/r04_dac_kernels_bs32_v3.txt 2>&1
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_dac_stage_parity_gpu.py tests/test_dac_gpu.py -m gpu -q -k "stage or fused or ragged" 2>&1 | tail -15 ) > gpurun_out/r04_gputest4.txt
tail -5 gpurun_out/r04_gputest4.txt; cat gpurun_out/r04_probes4.txt | cut -c1-220
