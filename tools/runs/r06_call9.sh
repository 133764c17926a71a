#!/bin/bash
# round 6, GPU call 9: kernel table of the time-to-first-token path at 32 utterances on the LDS-DMA GEMM tree
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/pp32; PROF_B=32 PROF_N=8 timeout 400 rocprofv3 --kernel-trace -d /tmp/pp32 -o p -- python $R/tools/prof_prefill.py > /tmp/pp32.log 2>&1
python $R/tools/prof_report.py $(find /tmp/pp32 -name '*.db' | head -1) 34 > $R/gpurun_out/r06_prefill_kernels_bs32_v2.txt 2>&1
cat $R/gpurun_out/r06_prefill_kernels_bs32_v2.txt | cut -c1-170
