#!/bin/bash
# round-2 GPU probe 7: hunt the intermittent streaming hang (stress with watchdog), then the suite with a per-test timeout, then bench
O=gpurun_out/r02g; mkdir -p $O; export TMPDIR=/tmp
timeout 400 python tools/stress_streamer.py 25 > $O/stress.log 2>&1; echo "stress rc=$?" >> $O/stress.log
timeout 900 python -m pytest tests -m gpu -q --timeout 240 --durations=4 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 200 python tools/step_probe2.py 32 mini >> $O/steps.log 2>&1
timeout 600 python bench.py --steps 3 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" >> $O/bench_n1.err
PTTS_DUMMY=1 timeout 400 python bench.py --bs 32 --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_bs32.json 2> $O/bench_bs32.err
tail -30 $O/stress.log; tail -15 $O/pytest.log; grep step_probe $O/steps.log; cat $O/bench_n1.json; tail -2 $O/bench_n1.err; cat $O/bench_bs32.json
