#!/bin/bash
# round-2 GPU probe 9: suite + smoke on the final build (async fold, 1-wave cross-attention at batch > 4), TTFT / bs=32 check
O=gpurun_out/r02i; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --durations=4 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 200 python tools/step_probe2.py 32 mini >> $O/steps.log 2>&1
timeout 200 python tools/step_probe2.py 8 mini >> $O/steps.log 2>&1
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" >> $O/bench_n1.err
tail -8 $O/pytest.log; tail -3 $O/smoke.log; grep step_probe $O/steps.log; cat $O/bench_n1.json; tail -2 $O/bench_n1.err
