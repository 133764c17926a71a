#!/bin/bash
# round 6, GPU call 32: (a) block order of the transposed convs (conv_lds_kernel a.order; dev-knob build PTTS_DAC_UP_ORDER=0 = rounds 2-5 order) A/B + bit
# identity + the codec suites on the product library; (b) tools/gemm_probe with one-workgroup-per-CU tiles, deep rings and the half-stage pipeline;
# (c) the vendor library on the same shapes, graph-captured
cd "$GRAFT_REPO_ROOT" || exit 1
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
P=tools/variants/dacdev/cabi_probe_dacdev
{
for i in 1 2; do
for o in 0 auto 1 2; do
  if [ $o = auto ]; then timeout 300 $P dac 32 tag=order_auto 2>&1 | grep -v "$F" | cut -c1-200
  else PTTS_DAC_UP_ORDER=$o timeout 300 $P dac 32 tag=order$o 2>&1 | grep -v "$F" | cut -c1-200; fi
done
done
PTTS_DAC_UP_ORDER=0 timeout 300 $P dac 1 tag=order0 2>&1 | grep -v "$F" | cut -c1-200
timeout 300 $P dac 1 tag=order_auto 2>&1 | grep -v "$F" | cut -c1-200
PTTS_DAC_UP_ORDER=0 timeout 300 $P dac 4 tag=order0 2>&1 | grep -v "$F" | cut -c1-200
timeout 300 $P dac 4 tag=order_auto 2>&1 | grep -v "$F" | cut -c1-200
PTTS_DAC_XIN=1 timeout 300 $P dac 4 tag=order_auto_xin1 2>&1 | grep -v "$F" | cut -c1-200
PTTS_DAC_XIN=7 timeout 300 $P dac 4 tag=order_auto_xin7 2>&1 | grep -v "$F" | cut -c1-200
PTTS_DAC_XIN=1 timeout 300 $P dac 8 tag=order_auto_xin1 2>&1 | grep -v "$F" | cut -c1-200
PTTS_DAC_XIN=7 timeout 300 $P dac 8 tag=order_auto_xin7 2>&1 | grep -v "$F" | cut -c1-200
PTTS_DAC_UP_ORDER=0 $P dac 3 frames=300 dump=/tmp/o0.bin tag=dump0 2>&1 | grep -v "$F" | cut -c1-200
$P dac 3 frames=300 dump=/tmp/oa.bin tag=dump_auto 2>&1 | grep -v "$F" | cut -c1-200
PTTS_DAC_UP_ORDER=1 $P dac 3 frames=300 dump=/tmp/o1.bin tag=dump1 2>&1 | grep -v "$F" | cut -c1-200
PTTS_DAC_UP_ORDER=2 $P dac 3 frames=300 dump=/tmp/o2.bin tag=dump2 2>&1 | grep -v "$F" | cut -c1-200
$P cmp /tmp/o0.bin /tmp/oa.bin; $P cmp /tmp/o0.bin /tmp/o1.bin; $P cmp /tmp/o0.bin /tmp/o2.bin
} | tee gpurun_out/r06_dac_up_order_ab.txt
timeout 1500 python -m pytest tests/test_dac_gpu.py tests/test_dac_stage_parity_gpu.py -q -m gpu -x 2>&1 | grep -v "$F" | tail -6 | tee gpurun_out/r06_gputest_call32.txt
timeout 900 tools/gemm_probe 40 2>&1 | grep -v "$F" | tee gpurun_out/r06_gemm_probe_call32.txt | tail -5
timeout 600 python tools/vendor_gemm_compare.py 2>&1 | grep -v "$F" | tee gpurun_out/r06_vendor_gemm.txt
