#!/bin/bash
# round 4, GPU call 5: conv_lds with deep weight prefetch (timing + table), SQ wave-state counters of the DAC kernels, DAC tests
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for B in 1 32; do timeout 120 tools/cabi_probe dac $B tag=wd3_all; done
timeout 120 tools/cabi_probe dac 32 f32 tag=f32
} > gpurun_out/r04_probes5.txt 2>&1
cd /tmp
rm -rf /tmp/pd32; timeout 300 rocprofv3 --kernel-trace -d /tmp/pd32 -o p -- $GRAFT_REPO_ROOT/tools/cabi_probe dac 32 reps=3 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/prof_report.py $(find /tmp/pd32 -name '*.db' | head -1) 14 > $GRAFT_REPO_ROOT/gpurun_out/r04_dac_kernels_bs32_v4.txt 2>&1
rm -rf /tmp/pq32; timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d /tmp/pq32 -o p -- $GRAFT_REPO_ROOT/tools/cabi_probe dac 32 reps=2 > /tmp/pq32.log 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_sq_report.py $(find /tmp/pq32 -name '*.db' | head -1) 14 > $GRAFT_REPO_ROOT/gpurun_out/r04_pmc_dac_sq.txt 2>&1
tail -3 /tmp/pq32.log >> $GRAFT_REPO_ROOT/gpurun_out/r04_pmc_dac_sq.txt
rm -rf /tmp/pm32; timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_VALU --kernel-trace -d /tmp/pm32 -o p -- $GRAFT_REPO_ROOT/tools/cabi_probe dac 32 reps=2 > /tmp/pm32.log 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_mfma_report.py $(find /tmp/pm32 -name '*.db' | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r04_pmc_dac_mfma.txt 2>&1
tail -3 /tmp/pm32.log >> $GRAFT_REPO_ROOT/gpurun_out/r04_pmc_dac_mfma.txt
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_dac_stage_parity_gpu.py tests/test_dac_gpu.py -m gpu -q 2>&1 | tail -15 ) > gpurun_out/r04_gputest5.txt
tail -5 gpurun_out/r04_gputest5.txt; cat gpurun_out/r04_probes5.txt | cut -c1-200; cat gpurun_out/r04_pmc_dac_sq.txt | cut -c1-220; cat gpurun_out/r04_pmc_dac_mfma.txt | cut -c1-200
