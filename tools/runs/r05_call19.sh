#!/bin/bash
# round 5, GPU call 19: 5..8 utterances - the GEMV step (7 nodes per layer, 12-wave workgroups) against the MFMA strip path of the wider engines (PTTS_NO_GEMV=1)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
for B in 5 6 8; do
  timeout 120 tools/cabi_probe lm $B tag=gemv_step
  PTTS_NO_GEMV=1 timeout 120 tools/cabi_probe lm $B tag=mfma_strips
done
PTTS_NO_GEMV=1 timeout 120 tools/cabi_probe lm 4 tag=mfma_strips
timeout 120 tools/cabi_probe lm 12 tag=mfma_strips
} > gpurun_out/r05_probes19.txt 2>&1
cat gpurun_out/r05_probes19.txt | cut -c1-200
