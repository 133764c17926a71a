"""GPU probe: end-to-end generate() of bench.py's configuration at SMALL batch B as one engine (the batched GEMV step: 7-8 nodes per layer,
no cross-attention folding above one utterance) vs n independent sub-batches on n engines and HIP streams (model.decode_streams = n,
decode_streams_min_sub = 1): n = B runs every utterance on its own single-utterance engine (5 fused nodes per layer at Mini-v1).
python tools/streams_probe_small.py [mini|large|large_fp8] 2 4 8 -> profiles/r04_experiments.txt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
dev = torch.device("cuda:0")
args = sys.argv[1:]
which = "mini"
if args and args[0] in ("mini", "large", "large_fp8"):
    which, args = args[0], args[1:]
model = bench.build_model_on_device(dev, torch.bfloat16, "large" if which.startswith("large") else "mini")
if which == "large_fp8":
    model.enable_fp8_weights()
for B in [int(x) for x in (args or ["2", "4", "8"])]:
    for n in (1, 2, 4, 8):
        if n > B or B % n:
            continue
        model.decode_streams = n
        model.decode_streams_min_sub = 1
        dt = bench._timed_generate(model, B, dev)
        print(f"[streams_probe_small {which}] B={B} decode_streams={n} ({B // n} per engine): {dt * 1e3:.1f} ms per generate() = {B * bench.AUDIO_S / dt:.1f} audio-s/s", flush=True)
    model.decode_streams = 0
    model._engine = None  # drop the engines (KV arenas) before the next batch size
