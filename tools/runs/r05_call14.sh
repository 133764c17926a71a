#!/bin/bash
# round 5, GPU call 14: does the box let a process pin the GPU's performance level, and does the latency-bound step care?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
rocm-smi --showperflevel --showclocks 2>&1 | grep -v "^$" | head -30
timeout 120 tools/cabi_probe lm 1 tag=auto &
sleep 6; rocm-smi --showclocks 2>&1 | grep -E "sclk|mclk|fclk|socclk" | head -8; wait
rocm-smi --setperflevel high 2>&1 | tail -3
rocm-smi --showperflevel 2>&1 | grep -i perf
timeout 120 tools/cabi_probe lm 1 tag=perflevel_high &
sleep 6; rocm-smi --showclocks 2>&1 | grep -E "sclk|mclk|fclk|socclk" | head -8; wait
timeout 120 tools/cabi_probe lm 32 tag=perflevel_high
rocm-smi --setperflevel auto 2>&1 | tail -2
timeout 120 tools/cabi_probe lm 32 tag=auto
ls /sys/class/drm/card*/device/power_dpm_force_performance_level 2>&1 | head -3
cat /sys/class/drm/card*/device/power_dpm_force_performance_level 2>&1 | head -3
} > gpurun_out/r05_probes14.txt 2>&1
cat gpurun_out/r05_probes14.txt | cut -c1-220
