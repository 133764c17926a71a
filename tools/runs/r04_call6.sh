#!/bin/bash
# round 4, GPU call 6: whole GPU suite, the bench line, the rocprofv3 summary of the bench command, DAC timing of the final kernels
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > gpurun_out/r04_gputest6.txt
{
for B in 1 32; do timeout 120 tools/cabi_probe dac $B tag=final; done
for B in 1 8 32 128; do timeout 120 tools/cabi_probe lm $B tag=final; done
} > gpurun_out/r04_probes6.txt 2>&1
timeout 1200 python bench.py > gpurun_out/r04_bench6.json.log 2> gpurun_out/r04_bench6.err
cd /tmp
rm -rf /tmp/pb; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pb -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > /tmp/pb.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_report.py $(find /tmp/pb -name '*.db' | head -1) 24 1734 > $GRAFT_REPO_ROOT/gpurun_out/r04_bench_bs1_rocprof_summary.txt 2>&1
tail -2 /tmp/pb.log | cut -c1-600 >> $GRAFT_REPO_ROOT/gpurun_out/r04_bench_bs1_rocprof_summary.txt
cd $GRAFT_REPO_ROOT
tail -6 gpurun_out/r04_gputest6.txt; cat gpurun_out/r04_probes6.txt | cut -c1-160; tail -c 400 gpurun_out/r04_bench6.json.log
