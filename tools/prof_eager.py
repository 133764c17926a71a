"""PMC target: Mini-v1 bf16 bs=1 decode steps launched EAGERLY (ptts_step_forward + ptts_push_tokens, no hipGraph) —
rocprofv3 --pmc crashes on graph replays in this image. Same kernels, same arguments as the captured step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from quick_probe import rand_sd
from parler_tts_amd.engine import DecoderEngine
B = int(os.environ.get("PROF_B", "1")); steps = int(os.environ.get("PROF_STEPS", "20"))
P = int(os.environ.get("PROF_P", "32"))  # prompt positions: a long prompt puts the eager steps at the context bench.py times (P + 1 + steps)
dev = torch.device("cuda:0"); H, L, F, K, V = 1024, 24, 4096, 9, 1088
eng = DecoderEngine(hidden_size=H, num_layers=L, num_heads=16, ffn_dim=F, num_codebooks=K, vocab_size=V, max_positions=4096,
                    dtype=torch.bfloat16, max_batch=B, max_ctx=max(940, P + 72), max_enc=64, max_prompt=P + 8)
eng.load_state_dict(rand_sd(H, L, F, K, V, 4096, dev))
eng.set_gen_params(max_length=steps + 40, min_new_tokens=steps + 39)
eng.prefill(torch.randn(B, 64, H, device=dev), None, torch.randn(B, P, H, device=dev), None, sample=False)
tok = torch.randint(0, 1024, (B * K,), device=dev)
for i in range(steps):
    eng.push_tokens(tok); eng.step_forward()
torch.cuda.synchronize(); print("eager steps done", steps)
