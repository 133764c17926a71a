"""Profile target (rocprofv3 --kernel-trace --stats): N prefills of bench.py's bs=1 configuration (T5 graph + prompt embedding + HIP
prefill of 33 positions + sampler tail + fold), no decode steps - the kernels on the time-to-first-token path."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
dev = torch.device("cuda:0")
bs = int(os.environ.get("PROF_B", "1")); n = int(os.environ.get("PROF_N", "20"))
model = bench.build_model_on_device(dev, torch.bfloat16, "mini")
print(f"ttft p50 {bench.measure_ttft(model, bs, dev, reps=n):.2f} ms", flush=True)
