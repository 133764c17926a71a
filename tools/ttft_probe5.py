"""Round 5: time-to-first-token and its breakdown (bench.measure_ttft / measure_ttft_breakdown) at batch 1 and 32 on bench.py's Mini-v1
configuration, weights filled on the device. Environment A/B: PTTS_NO_NATIVE_T5=1 (stock transformers T5), PTTS_PREFILL_GRAPH=0 (eager prefill),
PTTS_T5_NO_GRAPH=1 (eager T5 launches)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "default"
batches = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 32]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
model = bench.build_model_on_device(dev, torch.bfloat16, "mini")
for bs in batches:
    t = bench.measure_ttft(model, bs, dev, reps=15)
    b = bench.measure_ttft_breakdown(model, bs, dev, reps=15)
    print(f"[ttft_probe5 {tag}] bs={bs}: p50 {t:.3f} ms | " + json.dumps(b), flush=True)
