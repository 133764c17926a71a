#!/bin/bash
# round 4, GPU call 29 (final records): whole GPU suite, the bench line (timed), rocprofv3 summary of the bench command, batch-32 step table, prefill policy confirmation
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 ) > gpurun_out/r04_gputest29.txt
( time timeout 1200 python bench.py > gpurun_out/r04_bench29.json.log 2> gpurun_out/r04_bench29.err ) 2> gpurun_out/r04_bench29.time
{
for A in "1" "2" "1 large" "1 ctx=100"; do
  for V in 0 2; do PTTS_MSPLIT_PREFILL=$V timeout 120 tools/cabi_probe lm $A tag=x | sed -e 's/.*prefill+first token/prefill+first token/' -e 's/; create.*//' -e "s/^/[$A] prefill policy=$V: /"; done
done
for B in 1 8 32 64 128; do timeout 120 tools/cabi_probe lm $B tag=final | cut -c1-75; done
} > gpurun_out/r04_probes29.txt 2>&1
cd /tmp
rm -rf /tmp/pb; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pb -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > /tmp/pb.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_report.py $(find /tmp/pb -name '*.db' | head -1) 24 1734 > $GRAFT_REPO_ROOT/gpurun_out/r04_bench_bs1_rocprof_summary_v3.txt 2>&1
rm -rf /tmp/pl32; PROF_B=32 PROF_STEPS=20 timeout 400 rocprofv3 --kernel-trace -d /tmp/pl32 -o p -- python $GRAFT_REPO_ROOT/tools/prof_step.py > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/prof_report.py $(find /tmp/pl32 -name '*.db' | head -1) 16 420 > $GRAFT_REPO_ROOT/gpurun_out/r04_step_bf16_bs32_v2.txt 2>&1
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/r04_gputest29.txt; cat gpurun_out/r04_bench29.time; cat gpurun_out/r04_probes29.txt | cut -c1-110; tail -c 200 gpurun_out/r04_bench29.json.log
