#!/bin/bash
# round 4, GPU call 21: what fences do the step graph's kernel packets carry, and do the HIP runtime's switches move the dependent-launch cost?
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "## dispatch headers of graph-replayed kernel packets (AMD_LOG_LEVEL=4, 2 layers)"
AMD_LOG_LEVEL=4 timeout 120 tools/cabi_probe lm 1 layers=2 tag=log 2> /tmp/amdlog.txt | cut -c1-100
grep -o "Dispatch Header = 0x[0-9a-f]* (type=[0-9]*, barrier=[0-9]*, acquire=[0-9]*, release=[0-9]*)" /tmp/amdlog.txt | sort | uniq -c | sort -rn | head -12
grep -c "Dispatch Header" /tmp/amdlog.txt
grep -i -m 12 "graph\|Direct Dispatch" /tmp/amdlog.txt | cut -c1-220
echo "## step time under runtime switches"
timeout 120 tools/cabi_probe lm 1 tag=default
for V in "DEBUG_CLR_SKIP_RELEASE_SCOPE=1" "AMD_OPT_FLUSH=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1" "DEBUG_HIP_KERNARG_COPY_OPT=0" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" "DEBUG_HIP_DYNAMIC_QUEUES=0" "AMD_DIRECT_DISPATCH=0" "GPU_FLUSH_ON_EXECUTION=1" "ROC_USE_FGS_KERNARG=0"; do
  env $V timeout 120 tools/cabi_probe lm 1 tag=$V 2>&1 | tail -1
done
timeout 120 tools/cabi_probe lm 1 tag=default
timeout 120 tools/cabi_probe lm 32 tag=default
for V in "DEBUG_CLR_SKIP_RELEASE_SCOPE=1" "AMD_OPT_FLUSH=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"; do
  env $V timeout 120 tools/cabi_probe lm 32 tag=$V 2>&1 | tail -1
done
} > gpurun_out/r04_probes21.txt 2>&1
cut -c1-200 gpurun_out/r04_probes21.txt | tail -45
