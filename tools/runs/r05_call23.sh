#!/bin/bash
# round 5, GPU call 23 (final tree): the whole bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python bench.py ) > gpurun_out/r05_bench23.json.log 2> gpurun_out/r05_bench23.err
tail -3 gpurun_out/r05_bench23.err; head -c 400 gpurun_out/r05_bench23.json.log; echo
