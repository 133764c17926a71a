#!/bin/bash
# round-2 GPU probe 10: final bench line (fold on the caller's stream) + A/B of the async fold on TTFT / TTFA
O=gpurun_out/r02j; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" >> $O/bench_n1.err
PTTS_FOLD_ASYNC=1 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_async.json 2> $O/bench_async.err
cat $O/bench_n1.json; tail -2 $O/bench_n1.err; python -c "
import json;j=json.load(open('$O/bench_async.json'));print('async fold: value',j['value'],'ttft',j['ttft_p50_ms'],'ttfa',j['streaming'])"
