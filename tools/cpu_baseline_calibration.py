"""CPU tool (needs /root/reference; build container only): times the REFERENCE's own ParlerTTSForCausalLM
(modeling_parler_tts.py:1824, under oracle/reference_shims.py, DynamicCache, SDPA) beside the oracle port on the same
host, threads and shapes (Mini-v1 fp32, bs=1, 64 description + 32 prompt tokens), so that bench.py's `cpu_baseline`
(kind "port", the only thing that can run on the GPU box) can be read against the reference itself.
    PYTHONDONTWRITEBYTECODE=1 python tools/cpu_baseline_calibration.py [threads] [steps] > profiles/r02_cpu_baseline_calibration.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from oracle import decoder_oracle as DO
from oracle import make_golden as MG
from oracle.reference_shims import import_reference


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else min(os.cpu_count() or 8, 8)
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    torch.set_num_threads(threads)
    from transformers.cache_utils import DynamicCache, EncoderDecoderCache

    ref = import_reference()
    spec = DO.MINI_V1
    sd = DO.make_decoder_weights(spec, seed=1234)
    g = torch.Generator().manual_seed(1)
    N, P = 64, 32
    enc = torch.randn(1, N, spec.hidden_size, generator=g)
    prompt = torch.randn(1, P, spec.hidden_size, generator=g) * 0.02
    ids0 = torch.full((9, 1), spec.bos_token_id, dtype=torch.long)
    res = {}
    with torch.no_grad():
        m = MG.build_reference_lm(ref, spec, sd)
        for name in ("reference", "port", "reference", "port"):  # interleaved twice: second pass is the one reported
            if name == "reference":
                cache = EncoderDecoderCache(DynamicCache(), DynamicCache())
                t0 = time.perf_counter()
                logits = MG.reference_forward(m, cache, ids0, enc, None, prompt, None, 0)
                tp = time.perf_counter() - t0
                past = P + 1
                t1 = time.perf_counter()
                for _ in range(steps):
                    nxt = logits[:, -1].argmax(-1, keepdim=True).clamp(max=1023)
                    logits = MG.reference_forward(m, cache, nxt, enc, None, prompt, None, past)
                    past += 1
                ts = (time.perf_counter() - t1) / steps
                last_ref = logits[:, -1].clone()
            else:
                orc = DO.DecoderOracle(spec, sd)
                t0 = time.perf_counter()
                logits = orc.forward(ids0, enc, None, prompt, None)
                tp = time.perf_counter() - t0
                t1 = time.perf_counter()
                for _ in range(steps):
                    nxt = logits[:, -1].argmax(-1, keepdim=True).clamp(max=1023)
                    logits = orc.forward(nxt)
                ts = (time.perf_counter() - t1) / steps
                last_port = logits[:, -1].clone()
            res[name] = (tp, ts)
    print(f"# host: {os.cpu_count()} logical CPUs, {threads} torch threads; Mini-v1 fp32 bs=1, N={N}, P={P}, {steps} cached greedy steps (context {P + 1}..{P + 1 + steps})")
    for name, (tp, ts) in res.items():
        print(f"{name:10s} prefill {tp * 1e3:8.1f} ms   cached step {ts * 1e3:7.2f} ms")
    r = res["port"][1] / res["reference"][1]
    print(f"port / reference (cached step): {r:.3f}   max |dlogit| after {steps} steps (same greedy path): {float((last_ref - last_port).abs().max()):.2e}")
    print("# reading: bench.py's cpu_baseline (oracle port) is within this factor of the reference's own CPU path on this host; "
          "a port/reference ratio > 1 means the GPU/CPU ratio bench.py prints OVERSTATES the speed-up over the reference by that factor")


if __name__ == "__main__":
    main()
