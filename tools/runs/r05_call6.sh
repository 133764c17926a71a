#!/bin/bash
# round 5, GPU call 6: prefill-sized GEMMs - XCD-aware tile order on the register-blocked kernel, the LDS-tiled kernel with staging two stages ahead
# (ext_vector registers: no scratch), time to the first token at 32 utterances; T5 + LM parity tests on the new kernels; node stamps record
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_t5_gpu.py tests/test_lm_gpu.py -m gpu -x -q -k "t5 or prefill or batch or block" 2>&1 | tail -5 ) > gpurun_out/r05_gputest6.txt
cat gpurun_out/r05_gputest6.txt
{
PTTS_GEMM_TILE=0 timeout 300 python tools/ttft_probe5.py block+xcd 32
PTTS_GEMM_TILE=0 PTTS_GEMM_XCD=0 timeout 300 python tools/ttft_probe5.py block_launch_order 32
timeout 300 python tools/ttft_probe5.py tile_auto+xcd 32
PTTS_GEMM_XCD=0 timeout 300 python tools/ttft_probe5.py tile_auto_launch_order 32
for t in 88 48 84 44; do PTTS_GEMM_TILE=$t timeout 300 python tools/ttft_probe5.py tile$t+xcd 32; done
} > gpurun_out/r05_probes6.txt 2>&1
grep ttft_probe5 gpurun_out/r05_probes6.txt | cut -c1-200
