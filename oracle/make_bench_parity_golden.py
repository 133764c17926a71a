"""TEST INFRASTRUCTURE ONLY - generates tests/golden/bench_parity_ids.npz: the greedy ids of the oracle's own free run on the
configuration bench.py measures (Mini-v1, 24 layers, 64 description + 32 prompt tokens, 868 passes; model / seeds / inputs built by
bench.py itself), so the GPU parity test does not have to spend a minute of GPU-box time on 868 SEQUENTIAL CPU passes every run:
it teacher-forces one BATCHED oracle forward on these ids (seconds), re-derives the arg-max margins from it, and checks that the
stored ids are that forward's own arg-max on every pass before the first unsafe margin (tests/test_bench_config_parity_gpu.py).

    python oracle/make_bench_parity_golden.py        # ~2 min on 8 cores; CPU only
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import decoder_oracle as DO  # noqa: E402


def main():
    import bench

    torch.set_num_threads(min(os.cpu_count() or 8, 16))
    dev = torch.device("cpu")
    model = bench.build_model(0, 1, dev, torch.float32)
    desc, prompt = bench.synthetic_batch(1, 0, dev)
    with torch.no_grad():
        enc = model._encode_description(desc, None).float()
        pr = model.embed_prompts(prompt).float()
    sd = {k: v.detach().float() for k, v in model.decoder.state_dict().items()}
    L = bench.NEW_TOKENS + 1
    t0 = time.time()
    ref = DO.sample_loop(DO.DecoderOracle(DO.MINI_V1, sd), enc, None, pr, None, DO.GenParams(max_length=L, min_new_tokens=bench.NEW_TOKENS), keep_logits=True)
    margins = []
    for lg in ref.step_logits:
        lg = lg.clone()
        lg[:, DO.MINI_V1.eos_token_id] = -float("inf")
        t2 = torch.topk(lg, 2, dim=-1)[0]
        margins.append(float((t2[:, 0] - t2[:, 1]).min()))
    out = os.path.join(ROOT, "tests", "golden", "bench_parity_ids.npz")
    np.savez_compressed(out, ids=ref.sequences.numpy().astype(np.int16), margins=np.asarray(margins, dtype=np.float32),
                        enc_abs_sum=np.float64(enc.abs().sum().item()), note="oracle/make_bench_parity_golden.py; description encoder run on the CPU")
    first = next((i for i, m in enumerate(margins) if m < 2e-4), len(margins))
    print(f"wrote {out}: ids {tuple(ref.sequences.shape)}, first pass with margin < 2e-4: {first}, oracle {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
