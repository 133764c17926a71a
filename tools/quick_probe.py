"""GPU probe (not a test): per-step latency of the Mini-v1 decode graph at a few batch sizes + DAC decode time."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from parler_tts_amd.engine import DecoderEngine, DacEngine

def rand_sd(H, L, F, K, V, maxpos, dev):
    g = torch.Generator(device=dev).manual_seed(1234)
    r = lambda *s: torch.randn(*s, device=dev, generator=g) * 0.02
    sd = {}
    p = "model.decoder."
    for k in range(K): sd[f"{p}embed_tokens.{k}.weight"] = r(V + 1, H)
    sd[f"{p}embed_positions.weights"] = r(maxpos, H)
    for i in range(L):
        lp = f"{p}layers.{i}."
        for att in ("self_attn", "encoder_attn"):
            for pr in ("q_proj", "k_proj", "v_proj", "out_proj"): sd[f"{lp}{att}.{pr}.weight"] = r(H, H)
            sd[f"{lp}{att}_layer_norm.weight"] = torch.ones(H, device=dev); sd[f"{lp}{att}_layer_norm.bias"] = torch.zeros(H, device=dev)
        sd[f"{lp}fc1.weight"] = r(F, H); sd[f"{lp}fc2.weight"] = r(H, F)
        sd[f"{lp}final_layer_norm.weight"] = torch.ones(H, device=dev); sd[f"{lp}final_layer_norm.bias"] = torch.zeros(H, device=dev)
    sd[f"{p}layer_norm.weight"] = torch.ones(H, device=dev); sd[f"{p}layer_norm.bias"] = torch.zeros(H, device=dev)
    for k in range(K): sd[f"lm_heads.{k}.weight"] = r(V, H)
    return sd

def main():
    dev = torch.device("cuda:0")
    H, L, F, K, V = 1024, 24, 4096, 9, 1088
    sd = rand_sd(H, L, F, K, V, 4096, dev)
    for dtype in (torch.bfloat16, torch.float32):
        for B in ((1, 8, 32) if dtype == torch.bfloat16 else (1,)):
            eng = DecoderEngine(hidden_size=H, num_layers=L, num_heads=16, ffn_dim=F, num_codebooks=K, vocab_size=V, max_positions=4096,
                                dtype=dtype, max_batch=B, max_ctx=940, max_enc=64, max_prompt=40)
            t0 = time.time(); eng.load_state_dict(sd); torch.cuda.synchronize(); tl = time.time() - t0
            eng.set_gen_params(max_length=869, min_new_tokens=868)
            enc = torch.randn(B, 64, H, device=dev); prompt = torch.randn(B, 32, H, device=dev)
            torch.cuda.synchronize(); t0 = time.time()
            eng.prefill(enc, None, prompt, None); torch.cuda.synchronize(); tp = time.time() - t0
            t0 = time.time(); eng.prefill(enc, None, prompt, None); torch.cuda.synchronize(); tp2 = time.time() - t0
            eng.decode_steps(20); torch.cuda.synchronize()
            t0 = time.time(); eng.decode_steps(400); torch.cuda.synchronize(); t1 = time.time() - t0
            t0 = time.time(); eng.decode_steps(400); torch.cuda.synchronize(); t2 = time.time() - t0
            cur, fin = eng.state()
            print(f"[lm] {str(dtype):15s} B={B:3d} load {tl:.1f}s prefill {tp*1e3:.1f}/{tp2*1e3:.1f} ms  step {t1/400*1e6:.1f} us (ctx~250) {t2/400*1e6:.1f} us (ctx~650) cur_len={cur}", flush=True)
            eng.close(); del eng
    # Large-v1 decoder shapes (helpers/model_init_scripts/init_large_model.py:25-43), bf16
    H2, L2, F2 = 1536, 30, 6144
    sd2 = rand_sd(H2, L2, F2, K, V, 4096, dev)
    for B in (1, 8):
        eng = DecoderEngine(hidden_size=H2, num_layers=L2, num_heads=24, ffn_dim=F2, num_codebooks=K, vocab_size=V, max_positions=4096,
                            dtype=torch.bfloat16, max_batch=B, max_ctx=940, max_enc=64, max_prompt=40)
        eng.load_state_dict(sd2); eng.set_gen_params(max_length=869, min_new_tokens=868)
        eng.prefill(torch.randn(B, 64, H2, device=dev), None, torch.randn(B, 32, H2, device=dev), None)
        eng.decode_steps(300); torch.cuda.synchronize()
        t0 = time.time(); eng.decode_steps(300); torch.cuda.synchronize(); t1 = time.time() - t0
        wb = (L2 * (6 * H2 * H2 + 2 * H2 * F2) + K * V * H2) * 2
        print(f"[lm-large] bf16 B={B}: step {t1/300*1e6:.1f} us (ctx~480) -> {wb/1e9/(t1/300)/1e3:.2f} TB/s of weights alone ({wb/1e6:.0f} MB/step)", flush=True)
        eng.close(); del eng
    # DAC full size
    from parler_tts_amd.synthetic import random_dac_state_dict
    dsd = {k: v.to(dev) for k, v in random_dac_state_dict().items()}
    for B, T in ((1, 860), (4, 860)):
        dac = DacEngine(max_batch=B, max_frames=T)
        dac.load_state_dict(dsd)
        codes = torch.randint(0, 1024, (B, 9, T), device=dev)
        dac.decode(codes); torch.cuda.synchronize()
        t0 = time.time(); w = dac.decode(codes); torch.cuda.synchronize(); td = time.time() - t0
        print(f"[dac] B={B} T={T}: {td*1e3:.1f} ms  -> {B*T*1.608e9/td/1e12:.1f} TFLOP/s fp32-MFMA, {B*T*512/44100/td:.0f} audio-s/s", flush=True)
        dac.close(); del dac

if __name__ == "__main__":
    main()
