"""GPU probe: time-to-first-audio of generate(streamer=...) per repetition (Mini-v1 bf16, play_steps 43), with the chunk arrival times."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import parler_tts_amd as P

dev = torch.device("cuda", 0)
model = bench.build_model(0, 1, dev, torch.bfloat16)
desc, prompt = bench.synthetic_batch(1, 0, dev)
play_steps = 43
kw = dict(input_ids=desc, prompt_input_ids=prompt, do_sample=False, max_new_tokens=3 * play_steps, min_new_tokens=3 * play_steps)
model.generate(**kw)  # engines, graphs
for i in range(8):
    st = P.ParlerTTSStreamer(model, device=dev, play_steps=play_steps)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = threading.Thread(target=model.generate, kwargs=dict(streamer=st, **kw)); th.start()
    arr = []
    for c in st:
        arr.append((round((time.perf_counter() - t0) * 1e3, 1), len(c)))
    th.join()
    print(os.environ.get("TAG", ""), "rep", i, arr, flush=True)
