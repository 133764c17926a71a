"""GPU probe: end-to-end generate() of bench.py's configuration at batch B as ONE engine vs n independent sub-batches on n engines and
HIP streams (model.decode_streams = n): at >= 64 utterances one half's bandwidth-bound self-attention can overlap the other half's
latency-bound GEMM chain. python tools/streams_probe.py 128 64 -> profiles/r04_experiments.txt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
dev = torch.device("cuda:0")
model = bench.build_model_on_device(dev, torch.bfloat16, "mini")
for B in [int(x) for x in (sys.argv[1:] or ["128"])]:
    for n in (1, 2, 4):
        if B // n < 16 or n > int(os.environ.get('STREAMS_PROBE_MAX_N', '4')):
            continue
        model.decode_streams = n
        model.decode_streams_min_sub = 16
        dt = bench._timed_generate(model, B, dev)
        print(f"[streams_probe] B={B} decode_streams={n}: {dt * 1e3:.1f} ms per generate() = {B * bench.AUDIO_S / dt:.1f} audio-s/s", flush=True)
    model.decode_streams = 0
    model._engine = None  # drop the engines (KV arenas) before the next batch size
