#!/bin/bash
# round 5, GPU call 1: native T5 encoder parity + the TTFT breakdown (native vs stock T5, prefill graph vs eager), then the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_t5_gpu.py -x -q > gpurun_out/r05_t5_tests1.txt 2>&1; echo "t5 tests rc=$?" >> gpurun_out/r05_t5_tests1.txt
tail -5 gpurun_out/r05_t5_tests1.txt
{
timeout 300 python tools/ttft_probe5.py native+pgraph
PTTS_PREFILL_GRAPH=0 timeout 300 python tools/ttft_probe5.py native+eager_prefill
PTTS_T5_NO_GRAPH=1 timeout 300 python tools/ttft_probe5.py native_eager_t5+pgraph 1
PTTS_NO_NATIVE_T5=1 PTTS_PREFILL_GRAPH=0 timeout 300 python tools/ttft_probe5.py stock_t5+eager_prefill
} > gpurun_out/r05_probes1.txt 2>&1
grep ttft_probe5 gpurun_out/r05_probes1.txt | cut -c1-600
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest1.txt 2>&1; echo "suite rc=$?" >> gpurun_out/r05_gputest1.txt
tail -8 gpurun_out/r05_gputest1.txt
