#!/bin/bash
# round 6, GPU call 27: LM / generate / bench-config parity tests on the lnproj + xattn preload tree (call 26 cut the summary line off) + the codec's PMC databases
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 2000 python -m pytest tests/test_lm_gpu.py tests/test_generate_gpu.py tests/test_bench_config_parity_gpu.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids" | tail -12 | tee gpurun_out/r06_gputest_call27.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/dac_trace -o p -- $GRAFT_REPO_ROOT/tools/cabi_probe dac 32 reps=2 > /dev/null 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do timeout 300 rocprofv3 --pmc $c -d /tmp/dac_$c -o p -- $GRAFT_REPO_ROOT/tools/cabi_probe dac 32 reps=2 > /tmp/dac_$c.log 2>&1; done )
mkdir -p gpurun_out/dacdb
cp $(find /tmp/dac_FETCH_SIZE -name "*.db" | head -1) gpurun_out/dacdb/fetch.db
cp $(find /tmp/dac_WRITE_SIZE -name "*.db" | head -1) gpurun_out/dacdb/write.db
cp $(find /tmp/dac_trace -name "*.db" | head -1) gpurun_out/dacdb/trace.db
ls -la gpurun_out/dacdb
