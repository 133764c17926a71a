#!/bin/bash
# round 4, GPU call 3: DAC per-stage parity, DAC residual-unit A/B (skip prefetch, 64-channel chunks), cold-vs-resident weights (layers=2/8/24), step tables
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_dac_stage_parity_gpu.py tests/test_dac_gpu.py tests/test_bench_config_parity_gpu.py -m gpu -q -k "dac or stage or ragged or fused" 2>&1 | tail -30 ) > gpurun_out/r04_gputest3.txt
{
for B in 1 32; do
  timeout 120 tools/cabi_probe dac $B tag=default
  PTTS_DAC_NO_SKIP_PREFETCH=1 timeout 120 tools/cabi_probe dac $B tag=no_skip_prefetch
  PTTS_DAC_KS1=1 timeout 120 tools/cabi_probe dac $B tag=ks1
  PTTS_DAC_KS1=1 PTTS_DAC_NO_SKIP_PREFETCH=1 timeout 120 tools/cabi_probe dac $B tag=ks1_no_skip_prefetch
done
for B in 1 32 128; do for L in 2 8 24; do timeout 120 tools/cabi_probe lm $B layers=$L tag=layers$L; done; done
} > gpurun_out/r04_probes3.txt 2>&1
cd /tmp
rm -rf /tmp/pd32; timeout 300 rocprofv3 --kernel-trace -d /tmp/pd32 -o p -- $GRAFT_REPO_ROOT/tools/cabi_probe dac 32 reps=3 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/prof_report.py $(find /tmp/pd32 -name '*.db' | head -1) 14 > $GRAFT_REPO_ROOT/gpurun_out/r04_dac_kernels_bs32_v2.txt 2>&1
for B in 32 128; do
  rm -rf /tmp/pl$B; PROF_B=$B PROF_STEPS=20 timeout 400 rocprofv3 --kernel-trace -d /tmp/pl$B -o p -- python $GRAFT_REPO_ROOT/tools/prof_step.py > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/prof_report.py $(find /tmp/pl$B -name '*.db' | head -1) 16 420 > $GRAFT_REPO_ROOT/gpurun_out/r04_step_bf16_bs${B}_v0.txt 2>&1
done
cd $GRAFT_REPO_ROOT
tail -8 gpurun_out/r04_gputest3.txt; cat gpurun_out/r04_probes3.txt | grep -v "^$" | cut -c1-150 | tail -20
