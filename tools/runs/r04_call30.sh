#!/bin/bash
# round 4, GPU call 30: whole GPU suite on the final tree (LM heads back on 64-row passes), batch-32 step
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -5 ) > gpurun_out/r04_gputest30.txt
{ timeout 120 tools/cabi_probe lm 32 tag=final2 | cut -c1-80; timeout 120 tools/cabi_probe lm 24 tag=final2 | cut -c1-80; } > gpurun_out/r04_probes30.txt 2>&1
cat gpurun_out/r04_gputest30.txt gpurun_out/r04_probes30.txt
