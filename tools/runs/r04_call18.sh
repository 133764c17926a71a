#!/bin/bash
# round 4, GPU call 18 (records of the tree with the fused single-utterance nodes): whole GPU suite, the bench line (timed), the rocprofv3 summary of the
# bench command, decode-step kernel tables at batch 1 / 32 / 128, step probes
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > gpurun_out/r04_gputest18.txt
( time timeout 1200 python bench.py > gpurun_out/r04_bench18.json.log 2> gpurun_out/r04_bench18.err ) 2> gpurun_out/r04_bench18.time
{
for B in 1 2 4 8 32 64 128; do timeout 120 tools/cabi_probe lm $B tag=call18; done
timeout 120 tools/cabi_probe lm 1 fp32 tag=call18
timeout 120 tools/cabi_probe lm 1 large tag=call18
timeout 120 tools/cabi_probe lm 1 large fp8 tag=call18
timeout 120 tools/cabi_probe lm 4 large fp8 tag=call18
} > gpurun_out/r04_probes18.txt 2>&1
cd /tmp
rm -rf /tmp/pb; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pb -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > /tmp/pb.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_report.py $(find /tmp/pb -name '*.db' | head -1) 24 1734 > $GRAFT_REPO_ROOT/gpurun_out/r04_bench_bs1_rocprof_summary_v2.txt 2>&1
tail -2 /tmp/pb.log | cut -c1-600 >> $GRAFT_REPO_ROOT/gpurun_out/r04_bench_bs1_rocprof_summary_v2.txt
find /tmp/pb -name '*stats*.csv' | head -3 | while read f; do echo "## $f"; head -14 "$f" | cut -c1-200; done > $GRAFT_REPO_ROOT/gpurun_out/r04_bench_bs1_rocprof_stats_csv_head.txt 2>&1
for B in 32 128; do
  rm -rf /tmp/pl$B; PROF_B=$B PROF_STEPS=20 timeout 400 rocprofv3 --kernel-trace -d /tmp/pl$B -o p -- python $GRAFT_REPO_ROOT/tools/prof_step.py > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/prof_report.py $(find /tmp/pl$B -name '*.db' | head -1) 16 420 > $GRAFT_REPO_ROOT/gpurun_out/r04_step_bf16_bs${B}_v1.txt 2>&1
done
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/r04_gputest18.txt; cat gpurun_out/r04_bench18.time; cat gpurun_out/r04_probes18.txt | cut -c1-110; tail -c 300 gpurun_out/r04_bench18.json.log
