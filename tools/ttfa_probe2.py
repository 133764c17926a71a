"""GPU probe: the same as ttfa_probe.py but in bench.py's order (a bs=32 generate() first, so the engines are the batch-32 ones)."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import parler_tts_amd as P

dev = torch.device("cuda", 0)
model = bench.build_model(0, 1, dev, torch.bfloat16)
d32, p32 = bench.synthetic_batch(32, 0, dev)
if os.environ.get("BS32", "1") == "1":
    model.generate(input_ids=d32, prompt_input_ids=p32, do_sample=False, max_new_tokens=bench.NEW_TOKENS, min_new_tokens=bench.NEW_TOKENS)
desc, prompt = bench.synthetic_batch(1, 0, dev)
play_steps = 43
kw = dict(input_ids=desc, prompt_input_ids=prompt, do_sample=False, max_new_tokens=3 * play_steps, min_new_tokens=3 * play_steps)
model.generate(**kw)
eng = model._engine
print("engine max_batch", eng.cfg.max_batch, "max_ctx", eng.cfg.max_ctx, flush=True)
t0 = time.perf_counter(); model.generate(**kw); torch.cuda.synchronize(); print("plain generate ms", round((time.perf_counter() - t0) * 1e3, 1), flush=True)
for i in range(5):
    st = P.ParlerTTSStreamer(model, device=dev, play_steps=play_steps)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = threading.Thread(target=model.generate, kwargs=dict(streamer=st, **kw)); th.start()
    arr = []
    for c in st:
        arr.append((round((time.perf_counter() - t0) * 1e3, 1), len(c)))
    th.join()
    print(os.environ.get("TAG", ""), "rep", i, arr, flush=True)
