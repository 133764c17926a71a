#!/bin/bash
# round 5, GPU call 20 (final tree): DAC parity tests + the whole bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_dac_stage_parity_gpu.py tests/test_dac_gpu.py tests/test_generate_gpu.py -m gpu -x -q 2>&1 | tail -4 ) > gpurun_out/r05_gputest20.txt
cat gpurun_out/r05_gputest20.txt
( time timeout 1500 python bench.py ) > gpurun_out/r05_bench20.json.log 2> gpurun_out/r05_bench20.err
tail -3 gpurun_out/r05_bench20.err; head -c 600 gpurun_out/r05_bench20.json.log; echo
