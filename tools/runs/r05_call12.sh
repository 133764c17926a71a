#!/bin/bash
# round 5, GPU call 12: kernarg preload on the nodes of the GEMV step (gemv_kernel, qkv_attn_kernel, xfold_attn_kernel take their first 56 argument bytes as
# scalars): step time with / without the compiler flag (tools/build_nopreload.sh), results bit-identical (dump + cmp), LM parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
{
timeout 120 tools/cabi_probe lm 1 tag=preload dump=/tmp/p1.bin steps=4
timeout 120 tools/nopre/cabi_probe_nopre lm 1 tag=no_preload dump=/tmp/n1.bin steps=4
tools/cabi_probe cmp /tmp/p1.bin /tmp/n1.bin | tail -3
for rep in 1 2; do
  timeout 120 tools/cabi_probe lm 1 tag=preload
  timeout 120 tools/nopre/cabi_probe_nopre lm 1 tag=no_preload
done
for B in 2 4 8; do
  timeout 120 tools/cabi_probe lm $B tag=preload
  timeout 120 tools/nopre/cabi_probe_nopre lm $B tag=no_preload
done
timeout 120 tools/cabi_probe lm 1 fp32 tag=preload
timeout 120 tools/nopre/cabi_probe_nopre lm 1 fp32 tag=no_preload
timeout 120 tools/cabi_probe lm 1 large tag=preload
timeout 120 tools/nopre/cabi_probe_nopre lm 1 large tag=no_preload
timeout 120 tools/cabi_probe lm 1 large fp8 tag=preload
timeout 120 tools/nopre/cabi_probe_nopre lm 1 large fp8 tag=no_preload
timeout 120 tools/cabi_probe lm 32 tag=preload
timeout 120 tools/nopre/cabi_probe_nopre lm 32 tag=no_preload
} > gpurun_out/r05_probes12.txt 2>&1
grep -E "cabi_probe|max|identical|differ" gpurun_out/r05_probes12.txt | cut -c1-230
( timeout 900 python -m pytest tests/test_lm_gpu.py tests/test_bench_config_parity_gpu.py -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/r05_gputest12.txt
cat gpurun_out/r05_gputest12.txt
