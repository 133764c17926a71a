#!/bin/bash
# round 5, GPU call 10: can a node's weights wait in the XCD's L2? (tools/prefetch_probe.hip: L2 / MALL residency across kernel boundaries, prefetch of the
# next node's weights by extra workgroups of the current node)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
timeout 200 tools/prefetch_probe
timeout 200 tools/prefetch_probe
} > gpurun_out/r05_probes10.txt 2>&1
cat gpurun_out/r05_probes10.txt
