#!/bin/bash
# round 4, GPU call 31 (last GPU minutes): rows per M pass of the WIDE projections (N >= 3072: QKV, fc1) separately from the narrow ones, 48..128 utterances
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for B in 64 128 96 48; do
  for W in 0 64 32 16; do
    if [ $W = 16 ] && [ $B != 48 ]; then continue; fi
    PTTS_MSPLIT_ROWS_WIDE=$W timeout 60 tools/cabi_probe lm $B tag=wide$W | cut -c1-78
  done
done
} > gpurun_out/r04_probes31.txt 2>&1
cat gpurun_out/r04_probes31.txt
