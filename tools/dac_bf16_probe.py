"""GPU probe (not a test): DAC decode time, exact-f32 MFMA vs bf16-operand mode, full 44 kHz stack, 860 frames."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from parler_tts_amd.engine import DacEngine
from parler_tts_amd.synthetic import random_dac_state_dict

dev = torch.device("cuda:0")
dsd = {k: v.to(dev) for k, v in random_dac_state_dict().items()}
for dt in (torch.float32, torch.bfloat16):
    for B, T in ((1, 860), (8, 860)):
        dac = DacEngine(max_batch=B, max_frames=T, compute_dtype=dt)
        dac.load_state_dict(dsd)
        codes = torch.randint(0, 1024, (B, 9, T), device=dev)
        dac.decode(codes); torch.cuda.synchronize()
        t0 = time.time(); dac.decode(codes); torch.cuda.synchronize(); td = time.time() - t0
        print(f"[dac {str(dt):15s}] B={B} T={T}: {td*1e3:.1f} ms -> {B*T*1.608e9/td/1e12:.1f} TFLOP/s, {B*T*512/44100/td:.0f} audio-s/s", flush=True)
        dac.close(); del dac
