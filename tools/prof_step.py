"""Profile target: Mini-v1 bf16 decode steps (prefill + N graph replays) for rocprofv3 --kernel-trace --stats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from quick_probe import rand_sd
from parler_tts_amd.engine import DecoderEngine

B = int(os.environ.get("PROF_B", "1")); steps = int(os.environ.get("PROF_STEPS", "100"))
dev = torch.device("cuda:0")
H, L, F, K, V = 1024, 24, 4096, 9, 1088
sd = rand_sd(H, L, F, K, V, 4096, dev)
eng = DecoderEngine(hidden_size=H, num_layers=L, num_heads=16, ffn_dim=F, num_codebooks=K, vocab_size=V, max_positions=4096,
                    dtype=torch.bfloat16, max_batch=B, max_ctx=940, max_enc=64, max_prompt=40)
eng.load_state_dict(sd)
eng.set_gen_params(max_length=869, min_new_tokens=868)
enc = torch.randn(B, 64, H, device=dev); prompt = torch.randn(B, 32, H, device=dev)
eng.prefill(enc, None, prompt, None)
eng.decode_steps(400)   # reach mid context (~430)
torch.cuda.synchronize()
import time; t0 = time.time()
eng.decode_steps(steps); torch.cuda.synchronize()
print(f"B={B} {steps} steps: {(time.time()-t0)/steps*1e6:.1f} us/step")
