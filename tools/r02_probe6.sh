#!/bin/bash
# round-2 GPU probe 6: folded cross-attention (correctness suite + step latency A/B + TTFT), chunk-halo test fix
O=gpurun_out/r02f; mkdir -p $O; export TMPDIR=/tmp
rm -f gpurun_out/r02_parity_bench_config.txt
timeout 1500 python -m pytest tests -m gpu -q --durations=4 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 200 python tools/step_probe2.py 1 mini_xfold >> $O/steps.log 2>&1
PTTS_NO_XFOLD=1 timeout 200 python tools/step_probe2.py 1 mini_noxfold >> $O/steps.log 2>&1
timeout 200 python tools/step_probe2.py 1 mini_xfold fp8 >> $O/steps.log 2>&1
timeout 300 python tools/step_probe2.py 1 large_xfold large >> $O/steps.log 2>&1
timeout 300 python tools/step_probe2.py 1 large_xfold large fp8 >> $O/steps.log 2>&1
timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" >> $O/bench_n1.err
PTTS_NO_XFOLD=1 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_n1_noxfold.json 2> $O/bench_n1_noxfold.err
tail -8 $O/pytest.log; grep step_probe $O/steps.log; cat gpurun_out/r02_parity_bench_config.txt; cat $O/bench_n1.json; tail -2 $O/bench_n1.err; python -c "
import json;j=json.load(open('$O/bench_n1_noxfold.json'));print('noxfold: value',j['value'],'ttft',j['ttft_p50_ms'],'us/step',j['roofline']['us_per_launch'])"
