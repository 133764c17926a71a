#!/bin/bash
# round 5, GPU call 11: weights waiting in the Infinity Cache - a concurrent prefetcher on a second stream paced by a progress counter (tools/mall_prefetch_probe.hip)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 150 tools/mall_prefetch_probe > gpurun_out/r05_probes11.txt 2>&1
echo "rc=$?" >> gpurun_out/r05_probes11.txt
cat gpurun_out/r05_probes11.txt
