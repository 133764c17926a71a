#!/bin/bash
# round 4, GPU call 12: DAC decode of 32 utterances as 1 / 2 / 4 concurrent sub-batches on as many streams
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/dac_streams_probe.py > gpurun_out/r04_probes12.txt 2>&1
grep dac_streams gpurun_out/r04_probes12.txt
