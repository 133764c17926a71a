#!/bin/bash
# round 6, GPU call 19: gemm_strip_kernel with preloaded scalar arguments + hot tail fields in two cache lines: step times, stamps, LM parity subset
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "16" "32" "64" "128" "32 fp8" "32 large"; do timeout 300 tools/cabi_probe lm $cfg tag=strip_preload 2>&1 | grep -v "amdgpu.ids" | cut -c1-110; done
timeout 300 tools/stamps/cabi_probe_stamps lm 32 tag=stamps 2>&1 | grep "node stamps\] node\|per layer" | cut -c1-60,150-420
timeout 1200 python -m pytest tests/test_lm_gpu.py tests/test_t5_gpu.py -x -q -m gpu --durations=5 2>&1 | tail -12
