#!/bin/bash
# round 6, GPU call 47: whole GPU suite on the LDS-prefetch attention kernels + kernel table of the TTFT path at 32 utterances
R=$GRAFT_REPO_ROOT
cd $R || exit 1
export TMPDIR=/tmp
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | tail -12 > gpurun_out/r06_gputest_call47.txt; cat gpurun_out/r06_gputest_call47.txt
cd /tmp
rm -rf /tmp/pp32; PROF_B=32 PROF_N=8 timeout 400 rocprofv3 --kernel-trace -d /tmp/pp32 -o p -- python $R/tools/prof_prefill.py > /tmp/pp32.log 2>&1
python $R/tools/prof_report.py $(find /tmp/pp32 -name '*.db' | head -1) 30 > $R/gpurun_out/r06_prefill_kernels_bs32_v6.txt 2>&1
cd $R
head -16 gpurun_out/r06_prefill_kernels_bs32_v6.txt | cut -c1-170
