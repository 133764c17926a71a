"""GPU probe (not a test): decode-step latency (HIP-graph replays) of Mini-v1 / Large-v1 decoder shapes at batch B.
    python tools/step_probe2.py B tag [large] [fp8] [fp32]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from parler_tts_amd.engine import DecoderEngine
from quick_probe import rand_sd

B = int(sys.argv[1]); tag = sys.argv[2]; flags = set(sys.argv[3:])
dev = torch.device("cuda:0")
H, L, F, nh = (1536, 30, 6144, 24) if "large" in flags else (1024, 24, 4096, 16)
K, V = 9, 1088
sd = rand_sd(H, L, F, K, V, 4096, dev)
dtype = torch.float32 if "fp32" in flags else torch.bfloat16
eng = DecoderEngine(hidden_size=H, num_layers=L, num_heads=nh, ffn_dim=F, num_codebooks=K, vocab_size=V, max_positions=4096,
                    dtype=dtype, max_batch=B, max_ctx=940, max_enc=64, max_prompt=40, weights_fp8="fp8" in flags)
eng.load_state_dict(sd); eng.set_gen_params(max_length=869, min_new_tokens=868)
eng.prefill(torch.randn(B, 64, H, device=dev), None, torch.randn(B, 32, H, device=dev), None)
eng.decode_steps(50); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.time(); eng.decode_steps(250); torch.cuda.synchronize(); ts.append((time.time() - t0) / 250 * 1e6)
wb = (L * (6 * H * H + 2 * H * F) + K * V * H) * (1 if "fp8" in flags and B <= 4 else (4 if "fp32" in flags else 2))
print(f"[step_probe2 {tag} {' '.join(sorted(flags))}] B={B}: " + " ".join(f"{t:.1f}" for t in ts) + f" us/step  (weights {wb / 1e6:.0f} MB/step -> {wb / 1e6 / ts[1]:.2f} TB/s at the middle reading)", flush=True)
