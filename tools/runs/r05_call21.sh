#!/bin/bash
# round 5, GPU call 21 (final tree): the whole GPU suite + smoke()
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest21.txt 2>&1; echo "suite rc=$?" >> gpurun_out/r05_gputest21.txt
grep -E "passed|failed|error" gpurun_out/r05_gputest21.txt | tail -3; tail -1 gpurun_out/r05_gputest21.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke21.txt 2>&1; tail -1 gpurun_out/r05_smoke21.txt
