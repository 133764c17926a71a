#!/bin/bash
# round 6, GPU call 31: residual units on the fp32 stream (XIN; DESIGN section 8 open item 2): bit-identity + stage parity tests, A/B per width mask on one
# box (cabi_probe dac 32 / 1), per-kernel trace of mask 0 vs 7; the vendor library on the TTFT GEMM shapes (reference point for item 2's targets)
cd "$GRAFT_REPO_ROOT" || exit 1
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 1500 python -m pytest tests/test_dac_gpu.py tests/test_dac_stage_parity_gpu.py -q -m gpu -x 2>&1 | grep -v "$F" | tail -15 | tee gpurun_out/r06_gputest_call31.txt
{
for i in 1 2; do
for m in 0 1 2 4 3 7; do
  PTTS_DAC_XIN=$m timeout 300 tools/cabi_probe dac 32 tag=xin$m 2>&1 | grep -v "$F" | cut -c1-200
done
done
for m in 0 1 3 7; do
  PTTS_DAC_XIN=$m timeout 300 tools/cabi_probe dac 1 tag=xin$m 2>&1 | grep -v "$F" | cut -c1-200
done
PTTS_DAC_XIN=0 tools/cabi_probe dac 2 frames=300 dump=/tmp/x0.bin tag=dump0 2>&1 | grep -v "$F" | cut -c1-200
PTTS_DAC_XIN=7 tools/cabi_probe dac 2 frames=300 dump=/tmp/x7.bin tag=dump7 2>&1 | grep -v "$F" | cut -c1-200
tools/cabi_probe cmp /tmp/x0.bin /tmp/x7.bin
} | tee gpurun_out/r06_dac_xin_ab.txt
cd /tmp && export TMPDIR=/tmp
for m in 0 7; do
  PTTS_DAC_XIN=$m rocprofv3 --kernel-trace --stats -d /tmp/prof_xin$m -o xin$m -- $GRAFT_REPO_ROOT/tools/cabi_probe dac 32 reps=3 > /dev/null 2>&1
  f=$(find /tmp/prof_xin$m -name '*kernel_stats.csv' | head -1)
  { echo "== PTTS_DAC_XIN=$m"; head -25 "$f" | cut -c1-220; } >> $GRAFT_REPO_ROOT/gpurun_out/r06_dac_xin_kernels.txt
done
cd "$GRAFT_REPO_ROOT"
timeout 600 python tools/vendor_gemm_compare.py 2>&1 | grep -v "$F" | tee gpurun_out/r06_vendor_gemm.txt
