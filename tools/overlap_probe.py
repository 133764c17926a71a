"""GPU probe: end-to-end generate() of bench.py's configuration at batch B with the codec decoded chunk by chunk on a side stream
while the token graph runs (model.overlap_codec = True, BASELINE configs[4] "streaming DAC decode") against the sequential default."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
dev = torch.device("cuda:0")
model = bench.build_model_on_device(dev, torch.bfloat16, "mini")
for B in [int(x) for x in (sys.argv[1:] or ["32", "128"])]:
    for ov in (False, True):
        model.overlap_codec = ov
        dt = bench._timed_generate(model, B, dev)
        print(f"[overlap_probe] B={B} overlap_codec={ov}: {dt * 1e3:.1f} ms per generate() = {B * bench.AUDIO_S / dt:.1f} audio-s/s", flush=True)
