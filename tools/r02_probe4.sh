#!/bin/bash
# round-2 GPU probe 4: full suite (batch 2..4 GEMV, fp8 weights, streaming fix), step latencies Mini / Large / fp8, Large bench lines, PMC traffic
O=gpurun_out/r02d; mkdir -p $O; export TMPDIR=/tmp
rm -f gpurun_out/r02_parity_bench_config.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
for B in 1 2 4 8; do timeout 200 python tools/step_probe2.py $B mini >> $O/steps.log 2>&1; done
PTTS_GEMV_ROWS=1 timeout 200 python tools/step_probe2.py 4 mini_mfma_at_4 >> $O/steps.log 2>&1
for B in 1 4; do timeout 200 python tools/step_probe2.py $B mini fp8 >> $O/steps.log 2>&1; done
for B in 1 4; do timeout 300 python tools/step_probe2.py $B large large >> $O/steps.log 2>&1; timeout 300 python tools/step_probe2.py $B large large fp8 >> $O/steps.log 2>&1; done
timeout 200 python tools/step_probe2.py 1 mini fp32 >> $O/steps.log 2>&1
timeout 900 python bench.py --model large --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_large_bf16.json 2> $O/bench_large_bf16.err; echo "rc=$?" >> $O/bench_large_bf16.err
timeout 900 python bench.py --model large --dtype fp8w --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_large_fp8w.json 2> $O/bench_large_fp8w.err; echo "rc=$?" >> $O/bench_large_fp8w.err
timeout 900 python bench.py --model large --dtype fp8w --bs 4 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_large_fp8w_bs4.json 2> $O/bench_large_fp8w_bs4.err; echo "rc=$?" >> $O/bench_large_fp8w_bs4.err
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  PROF_STEPS=24 timeout 600 rocprofv3 --pmc $C -d $GRAFT_REPO_ROOT/$O/pmc_$C -o p -- python $GRAFT_REPO_ROOT/tools/prof_eager.py > $GRAFT_REPO_ROOT/$O/pmc_$C.log 2>&1
done
cd $GRAFT_REPO_ROOT
F=$(find $O/pmc_FETCH_SIZE -name "*.db" 2>/dev/null | head -1); W=$(find $O/pmc_WRITE_SIZE -name "*.db" 2>/dev/null | head -1)
if [ -n "$F" ] && [ -n "$W" ]; then python tools/pmc_report2.py $F $W 16 57 1 $O/r02_pmc_step_bs1.json > $O/r02_pmc_step_bs1.txt 2>&1; else echo "pmc passes produced no database" > $O/r02_pmc_step_bs1.txt; tail -5 $O/pmc_FETCH_SIZE.log >> $O/r02_pmc_step_bs1.txt; fi
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
tail -12 $O/pytest.log; grep step_probe $O/steps.log; cat $O/bench_large_bf16.json; tail -2 $O/bench_large_bf16.err; cat $O/bench_large_fp8w.json; tail -2 $O/bench_large_fp8w.err; cat $O/bench_large_fp8w_bs4.json; tail -2 $O/bench_large_fp8w_bs4.err; cat $O/r02_pmc_step_bs1.txt
