"""CPU study (oracle only, no GPU): what would an e4m3 self-attention KV cache cost in logits on the benchmarked model shape?

At batch 32 the self-attention KV stream is 1.66 GB of the 2.4 GB a decode step moves at mid context, and beyond
batch 32 it dominates; e4m3 K/V with one power-of-two scale per cached row (64 values) would halve it. That is a precision decision
before it is a kernel: this script measures it with the bf16-quantised oracle on Mini-v1 shapes (random-init weights, the bench's
seeds), teacher-forced on the fp32 oracle's greedy ids: max |dlogit| and arg-max agreement against the fp32 logits for
  (a) the bf16 engine numerics (bf16 weights / activations / KV),   (b) the same with the self-attention K and V rows stored as e4m3.
Caveat printed with the result: random-init attention is nearly uniform, so K errors are averaged away more than in a trained model.

    python tools/fp8_kv_study.py [steps=64] [layers=24]  > profiles/r02_fp8_kv_study.txt
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import decoder_oracle as DO  # noqa: E402
from oracle import fp8_oracle as FO  # noqa: E402


def q_rows_e4m3(t: torch.Tensor) -> torch.Tensor:
    """[..., 64] rows -> e4m3 with one power-of-two scale per row (what a 1-byte KV arena with a scale column would hold)"""
    flat = t.reshape(-1, t.shape[-1])
    return FO.quantize_rows(flat)[0].reshape(t.shape)


def run(spec, sd, enc, prompt, ids, steps, precision, kv8):
    orc = DO.DecoderOracle(spec, sd, precision=precision)
    outs = []

    def quantise_new(T):
        if not kv8:
            return
        for i in range(spec.num_hidden_layers):  # the positions this call appended
            orc.k_self[i][:, :, -T:, :] = q_rows_e4m3(orc.k_self[i][:, :, -T:, :])
            orc.v_self[i][:, :, -T:, :] = q_rows_e4m3(orc.v_self[i][:, :, -T:, :])

    with torch.no_grad():
        outs.append(orc.forward(ids[:, :1], enc, None, prompt, None)[:, -1])
        quantise_new(prompt.shape[1] + 1)
        for s in range(1, steps + 1):
            outs.append(orc.forward(ids[:, s:s + 1])[:, -1])
            quantise_new(1)
    return outs


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    layers = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    torch.set_num_threads(min(os.cpu_count() or 1, 8))
    spec = DO.DecoderSpec(**{**DO.MINI_V1.__dict__, "num_hidden_layers": layers})
    sd = DO.make_decoder_weights(spec, seed=1234)
    for k in range(spec.num_codebooks):
        sd[f"lm_heads.{k}.weight"][1024:] = 0.0
    g = torch.Generator().manual_seed(1)
    enc = torch.randn(1, 64, spec.hidden_size, generator=g)
    prompt = torch.randn(1, 32, spec.hidden_size, generator=g) * 0.02
    t0 = time.time()
    ref = DO.sample_loop(DO.DecoderOracle(spec, sd), enc, None, prompt, None, DO.GenParams(max_length=steps + 2, min_new_tokens=steps + 1), keep_logits=True)
    ids = DO.apply_delay_pattern_mask(ref.sequences, DO.build_delay_pattern_mask(ref.sequences[:, :1], spec.bos_token_id, spec.pad_token_id, steps + 2, spec.num_codebooks)[1])
    f32 = ref.step_logits[: steps + 1]
    print(f"# tools/fp8_kv_study.py: Mini-v1 shapes, {layers} layers, random-init weights (seed 1234), 64 description + 32 prompt positions, {steps + 1} passes,")
    print("# teacher-forced on the fp32 oracle's greedy ids; every figure is against the fp32 oracle's logits (|logit| ~ 0.3 here)")
    for name, kv8 in (("bf16 weights / activations / KV (the engine's numerics)", False), ("the same + self-attention K and V rows as e4m3 (power-of-two scale per row)", True)):
        outs = run(spec, sd, enc, prompt, ids, steps, "bf16", kv8)
        err = max(float((a - b).abs().max()) for a, b in zip(outs, f32))
        rms = (sum(float(((a - b) ** 2).mean()) for a, b in zip(outs, f32)) / len(outs)) ** 0.5
        agree = sum(int((a[:, :1024].argmax(-1) == b[:, :1024].argmax(-1)).sum()) for a, b in zip(outs, f32)) / (len(outs) * outs[0].shape[0])
        print(f"{name:82s} max |dlogit| {err:.3e}  rms {rms:.3e}  identical arg-max {100 * agree:.1f} %")
    print(f"# ({time.time() - t0:.0f} s on the CPU) Caveat: random-init attention is close to uniform over the context, which averages K / V rounding errors away;")
    print("# a trained checkpoint with peaked attention will be more sensitive to K. This bounds the cost from below, it does not settle the decision.")


if __name__ == "__main__":
    main()
