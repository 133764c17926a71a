#!/bin/bash
# round 6, GPU call 17: fc2 split-K machinery removed - LM / generate / bench-config parity suites + step times
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_lm_gpu.py tests/test_generate_gpu.py tests/test_bench_config_parity_gpu.py tests/test_native_abi.py -x -q -m gpu 2>&1 | tail -6
for cfg in "16" "32" "64" "128"; do timeout 300 tools/cabi_probe lm $cfg tag=unsplit_product 2>&1 | grep -v "amdgpu.ids" | cut -c1-110; done
