"""GPU probe: 44.1 kHz DAC decode time (bf16-operand mode) for 860 frames at batch 1 / 8 / 32 and for a streaming chunk,
plus the RMS difference to the direct (L1/L2-operand) kernel selected with PTTS_DAC_NO_LDS=1 in a second process."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from parler_tts_amd.engine import DacEngine
from parler_tts_amd.synthetic import random_dac_state_dict

tag = "direct" if os.environ.get("PTTS_DAC_NO_LDS") else "lds"
dsd = {k: v.cuda() for k, v in random_dac_state_dict(seed=4321).items()}
T = 860
for B in [int(x) for x in os.environ.get("PROBE_B", "1,8,32").split(",")]:
    dac = DacEngine(max_batch=B, max_frames=T, compute_dtype=torch.bfloat16)
    dac.load_state_dict(dsd)
    codes = torch.randint(0, 1024, (B, 9, T), generator=torch.Generator().manual_seed(7)).cuda()
    wav = dac.decode(codes)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5 if B > 1 else 20
    e0.record()
    for _ in range(n):
        wav = dac.decode(codes)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"[{tag}] B={B} T={T}: {ms:.3f} ms  ({1.608e9 * T * B / ms / 1e9:.0f} TFLOP/s)", flush=True)
    if B == 1:
        torch.save(wav.cpu(), os.path.join(ROOT, "gpurun_out", f"dac_probe_{tag}.pt"))
        c47 = codes[:, :, :47].contiguous()
        dac.decode(c47); torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            dac.decode(c47)
        e1.record(); torch.cuda.synchronize()
        print(f"[{tag}] B=1 T=47 (streaming chunk): {e0.elapsed_time(e1) / 20:.3f} ms", flush=True)
    del dac
other = os.path.join(ROOT, "gpurun_out", "dac_probe_direct.pt")
if tag == "lds" and os.path.exists(other):
    a, b = torch.load(os.path.join(ROOT, "gpurun_out", "dac_probe_lds.pt")), torch.load(other)
    print(f"lds vs direct: RMS diff {float((a - b).pow(2).mean().sqrt()):.3e}, signal RMS {float(b.pow(2).mean().sqrt()):.3e}", flush=True)
