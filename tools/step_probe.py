"""GPU probe (not a test): Mini-v1 bf16 decode-step latency at batch B (argv[1], default 1); prints one line.
Used to A/B runtime environment knobs (HIP_FORCE_DEV_KERNARG, DEBUG_CLR_GRAPH_PACKET_CAPTURE, ...)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from parler_tts_amd.engine import DecoderEngine
from quick_probe import rand_sd

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
tag = sys.argv[2] if len(sys.argv) > 2 else ""
dev = torch.device("cuda:0")
H, L, F, K, V = 1024, 24, 4096, 9, 1088
sd = rand_sd(H, L, F, K, V, 4096, dev)
eng = DecoderEngine(hidden_size=H, num_layers=L, num_heads=16, ffn_dim=F, num_codebooks=K, vocab_size=V, max_positions=4096,
                    dtype=torch.bfloat16, max_batch=B, max_ctx=940, max_enc=64, max_prompt=40)
eng.load_state_dict(sd); eng.set_gen_params(max_length=869, min_new_tokens=868)
eng.prefill(torch.randn(B, 64, H, device=dev), None, torch.randn(B, 32, H, device=dev), None)
eng.decode_steps(50); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.time(); eng.decode_steps(250); torch.cuda.synchronize(); ts.append((time.time() - t0) / 250 * 1e6)
print(f"[step_probe {tag}] B={B}: " + " ".join(f"{t:.1f}" for t in ts) + " us/step", flush=True)
