#!/bin/bash
# round 6, GPU call 12: VERDICT r05 item 4 - two half-batches on CU-masked streams at 128 and 64 utterances
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python tools/cumask_probe.py 128 64 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r06_cumask_probe.txt
