#!/bin/bash
# round 5, GPU call 16: DAC epilogues without store-acknowledge waits (Snake parameters through LDS, one straight-line pass per (stream?, format) instance):
# decode times + the DAC parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
{
timeout 120 tools/cabi_probe dac 32 tag=epilogue_v2
timeout 120 tools/cabi_probe dac 1 tag=epilogue_v2
timeout 120 tools/cabi_probe dac 32 tag=epilogue_v2
PTTS_DAC_NO_FUSE_RES=1 timeout 120 tools/cabi_probe dac 32 tag=epilogue_v2_unfused
timeout 120 tools/cabi_probe dac 8 tag=epilogue_v2
} > gpurun_out/r05_probes16.txt 2>&1
cat gpurun_out/r05_probes16.txt | cut -c1-200
( timeout 900 python -m pytest tests/test_dac_stage_parity_gpu.py tests/test_dac_gpu.py -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/r05_gputest16.txt
cat gpurun_out/r05_gputest16.txt
