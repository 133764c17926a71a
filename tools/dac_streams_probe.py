"""GPU probe: one DAC decode of 32 x 860 frames on one engine / stream against the same utterances as n sub-batches on n engines and n HIP streams
(do two decodes' MFMA-heavy and traffic-heavy phases fill each other's bubbles?). python tools/dac_streams_probe.py -> profiles/r04_experiments.txt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from parler_tts_amd.engine import DacEngine
from parler_tts_amd.synthetic import random_dac_state_dict

dev = torch.device("cuda:0")
sd = {k: v.to(dev) for k, v in random_dac_state_dict(seed=4321).items()}
B, T = 32, 860
codes = torch.randint(0, 1024, (B, 9, T), device=dev)
for n in (1, 2, 4):
    engs = []
    for _ in range(n):
        e = DacEngine(max_batch=B // n, max_frames=T, device=dev, compute_dtype=torch.bfloat16)
        e.load_state_dict(sd)
        engs.append(e)
    streams = [torch.cuda.Stream(dev) for _ in range(n)]
    parts = [codes[i * (B // n):(i + 1) * (B // n)].contiguous() for i in range(n)]

    def run():
        for e, s, c in zip(engs, streams, parts):
            with torch.cuda.stream(s):
                e.decode(c)

    for _ in range(2):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"[dac_streams_probe] {B} x {T} frames as {n} sub-batch(es) on {n} stream(s): {dt * 1e3:.2f} ms", flush=True)
    for e in engs:
        e.close()
