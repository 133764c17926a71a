#!/bin/bash
# round-2 GPU probe 3: full suite (streaming, PRO_LNS, GEMV variants), bs=32 A/B, bench line incl. TTFA, PMC traffic at mid context, bs=32 profile
O=gpurun_out/r02c; mkdir -p $O; export TMPDIR=/tmp
rm -f gpurun_out/r02_parity_bench_config.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 200 python tools/step_probe.py 1 base >> $O/steps.log 2>&1
timeout 200 python tools/step_probe.py 32 lns >> $O/steps.log 2>&1
PTTS_NO_LNS=1 timeout 200 python tools/step_probe.py 32 prep >> $O/steps.log 2>&1
timeout 200 python tools/step_probe.py 16 lns >> $O/steps.log 2>&1
timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" >> $O/bench_n1.err
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  PROF_STEPS=440 timeout 900 rocprofv3 --pmc $C -d $GRAFT_REPO_ROOT/$O/pmc_$C -o p -- python $GRAFT_REPO_ROOT/tools/prof_eager.py > $GRAFT_REPO_ROOT/$O/pmc_$C.log 2>&1
done
PROF_B=32 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof32 -o p -- python $GRAFT_REPO_ROOT/tools/prof_step.py > $GRAFT_REPO_ROOT/$O/prof32.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_report2.py $(find $O/pmc_FETCH_SIZE -name "*.db" | head -1) $(find $O/pmc_WRITE_SIZE -name "*.db" | head -1) 20 473 1 $O/r02_pmc_step_bs1.json > $O/r02_pmc_step_bs1.txt 2>&1
python tools/prof_report.py $(find $O/prof32 -name "*.db" | head -1) 24 100 > $O/prof32_report.txt 2>&1
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/prof32
tail -12 $O/pytest.log; grep step_probe $O/steps.log; cat gpurun_out/r02_parity_bench_config.txt; cat $O/bench_n1.json; tail -2 $O/bench_n1.err; cat $O/r02_pmc_step_bs1.txt; head -16 $O/prof32_report.txt
