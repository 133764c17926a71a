#!/bin/bash
# round 6, GPU call 11: cross K/V of all layers in one launch; T5 norms folded into the LDS-DMA GEMM above 256 rows - parity + TTFT breakdown
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_t5_gpu.py -x -q 2>&1 | tail -8
timeout 900 python -m pytest tests/test_lm_gpu.py -x -q -k "prefill or voice_prompt or batch_sizes or grouped_query" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_generate_gpu.py -x -q 2>&1 | tail -4
timeout 300 python - <<'PY'
import torch, bench, json
dev = torch.device("cuda:0")
model = bench.build_model_on_device(dev, torch.bfloat16, "mini")
for bs in (1, 32):
    print(bs, json.dumps({k: v for k, v in bench.measure_ttft_breakdown(model, bs, dev, reps=9).items() if k.endswith("_ms")}), flush=True)
PY
