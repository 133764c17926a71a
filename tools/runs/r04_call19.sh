#!/bin/bash
# round 4, GPU call 19: several decode steps per graph launch (PTTS_GRAPH_STEPS): ids identical, then step time A/B
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_bench_config_parity_gpu.py -m gpu -q -x -k "several_steps_per_graph" 2>&1 | tail -6 ) > gpurun_out/r04_gputest19.txt
{
for G in 1 4 16 1 8; do PTTS_GRAPH_STEPS=$G timeout 120 tools/cabi_probe lm 1 tag=graph_steps$G; done
for G in 1 8; do PTTS_GRAPH_STEPS=$G timeout 120 tools/cabi_probe lm 32 tag=graph_steps$G; done
for G in 1 8; do PTTS_GRAPH_STEPS=$G timeout 120 tools/cabi_probe lm 128 tag=graph_steps$G; done
for G in 1 8; do PTTS_GRAPH_STEPS=$G timeout 120 tools/cabi_probe lm 1 large tag=graph_steps$G; done
} > gpurun_out/r04_probes19.txt 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r04_gputest19.txt | head; cat gpurun_out/r04_probes19.txt | cut -c1-120
