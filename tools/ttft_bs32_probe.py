"""GPU probe (not a test): Mini-v1 bf16 prefill time at batch 32 (33 positions per utterance, 64 description tokens)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from parler_tts_amd.engine import DecoderEngine
from quick_probe import rand_sd
dev = torch.device("cuda:0")
H, L, F, K, V = 1024, 24, 4096, 9, 1088
sd = rand_sd(H, L, F, K, V, 4096, dev)
for B in (4, 8, 16, 32):
    eng = DecoderEngine(hidden_size=H, num_layers=L, num_heads=16, ffn_dim=F, num_codebooks=K, vocab_size=V, max_positions=4096,
                        dtype=torch.bfloat16, max_batch=B, max_ctx=940, max_enc=64, max_prompt=40)
    eng.load_state_dict(sd); eng.set_gen_params(max_length=869, min_new_tokens=868)
    enc = torch.randn(B, 64, H, device=dev); prompt = torch.randn(B, 32, H, device=dev)
    eng.prefill(enc, None, prompt, None); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.time(); eng.prefill(enc, None, prompt, None); torch.cuda.synchronize(); ts.append((time.time() - t0) * 1e3)
    print(f"[prefill] B={B}: " + " ".join(f"{t:.2f}" for t in ts) + " ms", flush=True)
    eng.close(); del eng
