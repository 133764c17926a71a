"""GPU probe: p50 time-to-first-token of bench.py's configuration (bs=1 and 32) as bench.measure_ttft reports it, plus the time until the
whole prefill call (incl. the cross-attention fold and the step-graph pre-capture on the host) has drained."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
dev = torch.device("cuda:0")
model = bench.build_model_on_device(dev, torch.bfloat16, "mini")
for bs in (1, 32):
    t0 = time.perf_counter(); first = bench.measure_ttft(model, bs, dev, reps=1); t_first = time.perf_counter() - t0
    print(f"bs={bs}: ttft p50 {bench.measure_ttft(model, bs, dev, reps=15):.2f} ms (first measurement round incl. graph pre-capture: {t_first * 1e3:.0f} ms for 4 calls)", flush=True)
