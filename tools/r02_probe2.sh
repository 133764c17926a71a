#!/bin/bash
# round-2 GPU probe 2: full -m gpu suite (new bench-config parity tests, sampler, tail), attention launch A/B, launcher proof, bench line
O=gpurun_out/r02b; mkdir -p $O; export TMPDIR=/tmp
rm -f gpurun_out/r02_parity_bench_config.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 200 python tools/step_probe.py 1 base >> $O/steps.log 2>&1
PTTS_ATTN_SPLITS=2 timeout 200 python tools/step_probe.py 1 S2 >> $O/steps.log 2>&1
PTTS_ATTN_SPLITS=2 PTTS_ATTN_WAVES=8 timeout 200 python tools/step_probe.py 1 S2W8 >> $O/steps.log 2>&1
PTTS_ATTN_SPLITS=1 PTTS_ATTN_WAVES=16 timeout 200 python tools/step_probe.py 1 S1W16 >> $O/steps.log 2>&1
PTTS_CROSS_WAVES=4 timeout 200 python tools/step_probe.py 1 X4 >> $O/steps.log 2>&1
timeout 600 python bench.py --steps 3 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" >> $O/bench_n1.err
PTTS_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_n2_nccl.json 2> $O/bench_n2_nccl.err; echo "rc=$?" >> $O/bench_n2_nccl.err
PTTS_BENCH_SHARE_GPU=1 PTTS_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err; echo "rc=$?" >> $O/bench_n2_gloo.err
tail -15 $O/pytest.log; cat $O/steps.log | grep step_probe; cat gpurun_out/r02_parity_bench_config.txt; cat $O/bench_n1.json; tail -3 $O/bench_n1.err; cat $O/bench_n2_nccl.json; tail -5 $O/bench_n2_nccl.err; cat $O/bench_n2_gloo.json; tail -3 $O/bench_n2_gloo.err
