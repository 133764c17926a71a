"""GPU probe (not a test): upper bound of what L2-resident weights would buy per kernel. A 1-layer decoder (F=2048:
21 MB of bf16 weights, fits the 8 x 4 MB L2) replays with warm weights when loads are allowed to allocate in L2;
the 24-layer model never does. Run once per library variant: PTTS_LIB=tools/libptts_cached.so python tools/l2_probe.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from parler_tts_amd.engine import DecoderEngine
from quick_probe import rand_sd

def main():
    dev = torch.device("cuda:0")
    H, K, V = 1024, 1, 1088
    for F in (2048, 4096):
        for L in (1, 2, 24):
            sd = rand_sd(H, L, F, K, V, 4096, dev)
            eng = DecoderEngine(hidden_size=H, num_layers=L, num_heads=16, ffn_dim=F, num_codebooks=K, vocab_size=V, max_positions=4096,
                                dtype=torch.bfloat16, max_batch=1, max_ctx=940, max_enc=64, max_prompt=40)
            eng.load_state_dict(sd); eng.set_gen_params(max_length=869, min_new_tokens=868)
            eng.prefill(torch.randn(1, 64, H, device=dev), None, torch.randn(1, 32, H, device=dev), None)
            eng.decode_steps(50); torch.cuda.synchronize()
            t0 = time.time(); eng.decode_steps(400); torch.cuda.synchronize(); t1 = time.time() - t0
            nodes = 7 * L + 2
            print(f"[{os.environ.get('PTTS_LIB', 'default')}] F={F} L={L:2d}: step {t1/400*1e6:8.1f} us = {t1/400*1e6/nodes:.2f} us/node ({nodes} nodes)", flush=True)
            eng.close(); del eng, sd

if __name__ == "__main__":
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    main()
