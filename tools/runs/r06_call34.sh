#!/bin/bash
# round 6, GPU call 34: the codec's per-kernel HBM bytes with the XIN residual units (the committed table r06_pmc_dac_bs32.txt predates them; bench.py's
# dac.bf16_bs32.hbm quotes its total) + gemm_probe on the T5 q|k|v / wi shapes at 64 and 128 descriptions (do the one-workgroup-per-CU tiles hold over 2-4 rounds?)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/dac_trace -o p -- $GRAFT_REPO_ROOT/tools/cabi_probe dac 32 reps=2 > /dev/null 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do timeout 300 rocprofv3 --pmc $c -d /tmp/dac_$c -o p -- $GRAFT_REPO_ROOT/tools/cabi_probe dac 32 reps=2 > /tmp/dac_$c.log 2>&1; done )
python tools/pmc_dac_report.py $(find /tmp/dac_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/dac_WRITE_SIZE -name "*.db" | head -1) $(find /tmp/dac_trace -name "*.db" | head -1) 3 2>&1 | tee gpurun_out/r06_pmc_dac_bs32_xin.txt
timeout 300 tools/cabi_probe dac 32 tag=product 2>&1 | grep -v "$F" | cut -c1-200 | tee -a gpurun_out/r06_pmc_dac_bs32_xin.txt
{ timeout 600 tools/gemm_probe 40 "T5 q" 2>&1 | grep -v "$F"; timeout 600 tools/gemm_probe 40 "T5 wi" 2>&1 | grep -v "$F"; } | tee gpurun_out/r06_gemm_probe_call34.txt | tail -3
