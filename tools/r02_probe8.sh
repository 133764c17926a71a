#!/bin/bash
# round-2 GPU probe 8: rocprofv3 kernel-trace summaries of the bench command and of the step (bs=1 folded, bs=32), final bench line
O=gpurun_out/r02h; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/profb -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/profb.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof1 -o p -- python $GRAFT_REPO_ROOT/tools/prof_step.py > $GRAFT_REPO_ROOT/$O/prof1.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_report.py $(find $O/profb -name "*.db" | head -1) 24 1734 > $O/profb_report.txt 2>&1
python tools/prof_report.py $(find $O/prof1 -name "*.db" | head -1) 16 100 > $O/prof1_report.txt 2>&1
rm -rf $O/profb $O/prof1
timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" >> $O/bench_n1.err
tail -3 $O/profb.log; cat $O/profb_report.txt; head -14 $O/prof1_report.txt; tail -2 $O/prof1_report.txt; cat $O/bench_n1.json
