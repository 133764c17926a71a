#!/bin/bash
# round 4, GPU call 2: whole GPU suite (no -x), DAC epilogue A/B + per-kernel table, batch 32 / 128 step tables, bench line
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > gpurun_out/r04_gputest2.txt
{
for B in 1 32; do
  PTTS_DAC_EPI_DIRECT=1 timeout 120 tools/cabi_probe dac $B tag=epi_direct
  timeout 120 tools/cabi_probe dac $B tag=epi_lds
done
for B in 32 64 128; do timeout 120 tools/cabi_probe lm $B tag=default; done
PTTS_XATTN_G=8 timeout 120 tools/cabi_probe lm 64 tag=xattn_g8
PTTS_XATTN_G=2 timeout 120 tools/cabi_probe lm 64 tag=xattn_g2
} > gpurun_out/r04_probes2.txt 2>&1
cd /tmp
for B in 32; do
  rm -rf /tmp/pd$B; timeout 300 rocprofv3 --kernel-trace -d /tmp/pd$B -o p -- $GRAFT_REPO_ROOT/tools/cabi_probe dac $B reps=3 > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/prof_report.py $(find /tmp/pd$B -name '*.db' | head -1) 24 > $GRAFT_REPO_ROOT/gpurun_out/r04_dac_kernels_bs$B.txt 2>&1
done
for B in 32 128; do
  rm -rf /tmp/pl$B; timeout 300 rocprofv3 --kernel-trace -d /tmp/pl$B -o p -- $GRAFT_REPO_ROOT/tools/cabi_probe lm $B > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/prof_report.py $(find /tmp/pl$B -name '*.db' | head -1) 16 750 > $GRAFT_REPO_ROOT/gpurun_out/r04_step_bf16_bs${B}_v0.txt 2>&1
done
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/r04_bench2.json.log 2> gpurun_out/r04_bench2.err
tail -12 gpurun_out/r04_gputest2.txt; cat gpurun_out/r04_probes2.txt | grep -v "^$" | tail -20; tail -c 600 gpurun_out/r04_bench2.json.log
