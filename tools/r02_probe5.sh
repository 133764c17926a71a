#!/bin/bash
# round-2 GPU probe 5: full suite, step latencies after the kernarg / exp2 changes, bench line
O=gpurun_out/r02e; mkdir -p $O; export TMPDIR=/tmp
rm -f gpurun_out/r02_parity_bench_config.txt
timeout 1500 python -m pytest tests -m gpu -q --durations=4 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
for B in 1 2 4; do timeout 200 python tools/step_probe2.py $B mini >> $O/steps.log 2>&1; done
timeout 200 python tools/step_probe2.py 1 mini fp8 >> $O/steps.log 2>&1
timeout 300 python tools/step_probe2.py 1 large large >> $O/steps.log 2>&1; timeout 300 python tools/step_probe2.py 1 large large fp8 >> $O/steps.log 2>&1
timeout 200 python tools/step_probe2.py 32 mini >> $O/steps.log 2>&1
timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" >> $O/bench_n1.err
tail -8 $O/pytest.log; grep step_probe $O/steps.log; cat gpurun_out/r02_parity_bench_config.txt; cat $O/bench_n1.json; tail -2 $O/bench_n1.err
