#!/bin/bash
# round 5, GPU call 4: s_memtime stamps of the single-utterance step (measurement build), then the whole bench line of this tree
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
{
timeout 120 tools/stamps/cabi_probe_stamps lm 1 tag=stamps
timeout 120 tools/stamps/cabi_probe_stamps lm 1 tag=stamps_again
timeout 120 tools/cabi_probe lm 1 tag=product_library
} > gpurun_out/r05_node_stamps.txt 2>&1
cut -c1-330 gpurun_out/r05_node_stamps.txt
( time timeout 1500 python bench.py ) > gpurun_out/r05_bench4.json.log 2> gpurun_out/r05_bench4.err
tail -3 gpurun_out/r05_bench4.err; tail -c 6000 gpurun_out/r05_bench4.json.log
