#!/bin/bash
# round 5, GPU call 7: T5LayerNorm folded into the GEMMs (<= 256 rows): parity + time to the first token with / without; step times after the strip-kernel change
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_t5_gpu.py -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/r05_gputest7.txt
cat gpurun_out/r05_gputest7.txt
{
timeout 300 python tools/ttft_probe5.py t5_fold 1
PTTS_T5_NO_FOLD=1 timeout 300 python tools/ttft_probe5.py t5_rows_prep 1
for B in 1 32 128; do timeout 120 tools/cabi_probe lm $B tag=after_strip_change; done
} > gpurun_out/r05_probes7.txt 2>&1
grep -E "ttft_probe5|cabi_probe" gpurun_out/r05_probes7.txt | cut -c1-200
( timeout 900 python -m pytest tests/test_lm_gpu.py tests/test_bench_config_parity_gpu.py -m gpu -x -q 2>&1 | tail -4 ) >> gpurun_out/r05_gputest7.txt
tail -4 gpurun_out/r05_gputest7.txt
