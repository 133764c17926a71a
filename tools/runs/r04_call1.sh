#!/bin/bash
# round 4, GPU call 1: full GPU suite (bf16 DAC oracle tolerances), xattn G variants, epilogue probe, stream-split at 128 / 64, bench line with `dac`
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r04_gputest1.txt
for B in 32 128; do for g in 8 4 2; do PTTS_XATTN_G=$g timeout 120 tools/cabi_probe lm $B tag=xattn_g$g; done; done > gpurun_out/r04_probes1.txt 2>&1
timeout 120 tools/epilogue_probe 32 > gpurun_out/r04_epilogue_probe.txt 2>&1
timeout 600 python tools/streams_probe.py 128 64 >> gpurun_out/r04_probes1.txt 2>&1
timeout 900 python bench.py > gpurun_out/r04_bench1.json.log 2> gpurun_out/r04_bench1.err
tail -5 gpurun_out/r04_gputest1.txt; cat gpurun_out/r04_probes1.txt | grep -v "^$" | tail -20; tail -c 1500 gpurun_out/r04_bench1.json.log
