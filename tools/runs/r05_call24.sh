#!/bin/bash
# round 5, GPU call 24 (final tree incl. the K / V append in the QKV node): the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest24.txt 2>&1; echo "suite rc=$?" >> gpurun_out/r05_gputest24.txt
grep -E "passed|failed|error" gpurun_out/r05_gputest24.txt | tail -3; tail -1 gpurun_out/r05_gputest24.txt
