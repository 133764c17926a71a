"""TTFT breakdown on the bench model: description encoder (stock PyTorch-ROCm T5) vs HIP prefill (+ first tail)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
dev = torch.device("cuda:0")
model = bench.build_model(0, 1, dev, torch.bfloat16)
for bs in (1, 32):
    desc, prompt = bench.synthetic_batch(bs, 0, dev)
    eng = model._get_engine(bs, bench.N_DESC, bench.N_PROMPT, bench.NEW_TOKENS + 1)
    eng.set_gen_params(max_length=bench.NEW_TOKENS + 1, min_new_tokens=bench.NEW_TOKENS)
    te, tp = [], []
    for i in range(13):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        enc = model._encode_description(desc, None).float(); pr = model.embed_prompts(prompt).float()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        eng.prefill(enc, None, pr, None, sample=True)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if i >= 3: te.append(t1 - t0); tp.append(t2 - t1)
    te.sort(); tp.sort()
    print(f"bs={bs}: T5 encoder + prompt embed p50 {te[5]*1e3:.2f} ms | HIP prefill ({bench.N_PROMPT+1} positions) + first tail p50 {tp[5]*1e3:.2f} ms", flush=True)
