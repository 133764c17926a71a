#!/bin/bash
# round 4, GPU call 10 (final records): whole GPU suite, the bench line (timed), the 2-rank functional run of the launcher (per-rank TTFT), DAC kernel table at batch 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > gpurun_out/r04_gputest10.txt
( time timeout 1200 python bench.py > gpurun_out/r04_bench_final.json.log 2> gpurun_out/r04_bench_final.err ) 2> gpurun_out/r04_bench_final.time
( PTTS_BENCH_SHARE_GPU=1 PTTS_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -3 | cut -c1-3000 ) > gpurun_out/r04_launcher_2rank_functional.txt
cd /tmp
rm -rf /tmp/pd1; timeout 300 rocprofv3 --kernel-trace -d /tmp/pd1 -o p -- $GRAFT_REPO_ROOT/tools/cabi_probe dac 1 reps=10 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/prof_report.py $(find /tmp/pd1 -name '*.db' | head -1) 14 > $GRAFT_REPO_ROOT/gpurun_out/r04_dac_kernels_bs1.txt 2>&1
rm -rf /tmp/pd32; timeout 300 rocprofv3 --kernel-trace -d /tmp/pd32 -o p -- $GRAFT_REPO_ROOT/tools/cabi_probe dac 32 reps=3 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/prof_report.py $(find /tmp/pd32 -name '*.db' | head -1) 14 > $GRAFT_REPO_ROOT/gpurun_out/r04_dac_kernels_bs32_final.txt 2>&1
cd $GRAFT_REPO_ROOT
grep -E "passed|failed" gpurun_out/r04_gputest10.txt; cat gpurun_out/r04_bench_final.time; tail -c 300 gpurun_out/r04_bench_final.json.log; echo; tail -c 600 gpurun_out/r04_launcher_2rank_functional.txt
