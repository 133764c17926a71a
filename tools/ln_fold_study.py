"""CPU study (oracle only, no GPU): LayerNorm folded into the consumer projection.

    W LN(x) = rstd * (W' x) - rstd * mean * c + d,   W' = W diag(gamma),  c_n = sum_k W'_nk,  d = W beta

With W', c, d prepared at load time the LN + projection node needs no prologue at all: every GEMV wave already holds the whole
activation row, so it can take mean / rstd from the row it loaded (two extra wave reductions) and correct its dot products in the
epilogue - no prologue wave, no LDS hand-off, no barrier (batch 1..4), and no per-workgroup re-normalisation of the 32-row block
(batch 9..32, PRO_LNS). What it changes is WHERE the bf16 rounding happens: raw x and W diag(gamma) are rounded instead of LN(x) and W.
This script measures that on the bench's model shape with non-trivial gamma / beta and a residual stream that is not zero-mean,
teacher-forced on the fp32 oracle's ids, every figure against the fp32 oracle:
  (a) the engine's current bf16 numerics, (b) the folded form in bf16, (c) the folded form in fp32 (the parity engine: cancellation only).

    python tools/ln_fold_study.py [steps=48] [layers=24] > profiles/r02_ln_fold_study.txt
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import decoder_oracle as DO  # noqa: E402


class _Normed:
    """what `_ln` hands to `_linear` in the folded oracle: the raw row and its statistics"""

    def __init__(self, x, mean, rstd, name):
        self.x, self.mean, self.rstd, self.name = x, mean, rstd, name


class FoldedOracle(DO.DecoderOracle):
    def _ln(self, x, name):
        mean = x.mean(-1, keepdim=True)
        rstd = torch.rsqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-5)
        return _Normed(x, mean, rstd, name)

    def _linear(self, x, name):
        if not isinstance(x, _Normed):
            return super()._linear(x, name)
        key = (x.name, name)
        cache = self.__dict__.setdefault("_fold", {})
        if key not in cache:  # prepared once, like a load-time weight transformation
            w32 = self.w32[name]
            wp = w32 * self.w[x.name + ".weight"][None, :]
            wp = DO.bf16_round(wp) if self.precision == "bf16" else wp
            cache[key] = (wp, wp.sum(-1), F.linear(self.w[x.name + ".bias"], w32))
        wp, c, d = cache[key]
        acc = F.linear(self._act(x.x), wp)
        return x.rstd * (acc - x.mean * c) + d


def make(spec, sd, precision, folded):
    cls = FoldedOracle if folded else DO.DecoderOracle
    orc = cls(spec, sd, precision=precision)
    if folded:
        orc.w32 = {k: v.detach().float() for k, v in sd.items()}  # the un-rounded matrices the fold starts from
    return orc


def run(orc, enc, prompt, ids, steps):
    outs = []
    with torch.no_grad():
        outs.append(orc.forward(ids[:, :1], enc, None, prompt, None)[:, -1])
        for s in range(1, steps + 1):
            outs.append(orc.forward(ids[:, s:s + 1])[:, -1])
    return outs


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    layers = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    torch.set_num_threads(min(os.cpu_count() or 1, 8))
    spec = DO.DecoderSpec(**{**DO.MINI_V1.__dict__, "num_hidden_layers": layers})
    base = DO.make_decoder_weights(spec, seed=1234)
    g = torch.Generator().manual_seed(1)
    enc = torch.randn(1, 64, spec.hidden_size, generator=g)
    prompt = torch.randn(1, 32, spec.hidden_size, generator=g) * 0.02
    print(f"# tools/ln_fold_study.py: Mini-v1 shapes, {layers} layers, random-init matrices (seed 1234), {steps + 1} passes teacher-forced on the fp32 oracle's greedy ids;")
    print("# every figure is against the fp32 oracle evaluating the SAME weights. 'offset' = constant added to every embedding row (mean / std of the residual")
    print("# stream at the first LayerNorm in brackets); gamma ~ U(0.5, 1.5) or with 8 channels at 8 (outlier scales), beta ~ N(0, 0.1)")
    t0 = time.time()
    for label, offset, outlier in (("gamma 1, beta 0, zero-mean stream (random init as is)", 0.0, None), ("gamma U(0.5,1.5), beta N(0,0.1), offset 0.05", 0.05, False),
                                   ("same, 8 gamma channels at 8.0, offset 0.2", 0.2, True)):
        sd = {k: v.clone() for k, v in base.items()}
        gg = torch.Generator().manual_seed(7)
        if outlier is not None:
            for k in sd:
                if k.endswith("layer_norm.weight"):
                    sd[k] = torch.rand(sd[k].shape, generator=gg) + 0.5
                    if outlier:
                        sd[k][torch.randperm(sd[k].numel(), generator=gg)[:8]] = 8.0
                elif k.endswith("layer_norm.bias"):
                    sd[k] = torch.randn(sd[k].shape, generator=gg) * 0.1
            for k in sd:
                if "embed_tokens" in k:
                    sd[k] = sd[k] + offset / spec.num_codebooks
        for k in range(spec.num_codebooks):
            sd[f"lm_heads.{k}.weight"][1024:] = 0.0
        ref_orc = DO.DecoderOracle(spec, sd)
        ref = DO.sample_loop(ref_orc, enc, None, prompt, None, DO.GenParams(max_length=steps + 2, min_new_tokens=steps + 1), keep_logits=True)
        ids = DO.apply_delay_pattern_mask(ref.sequences, DO.build_delay_pattern_mask(ref.sequences[:, :1], spec.bos_token_id, spec.pad_token_id, steps + 2,
                                                                                      spec.num_codebooks)[1])
        f32 = ref.step_logits[: steps + 1]
        x0 = ref_orc.embed(ids[:, :1])
        ratio = float(x0.mean().abs() / x0.std())
        scale = float(torch.stack(f32).abs().max())
        print(f"## {label}  [|mean| / std of the embedded row {ratio:.2f}; max |logit| {scale:.2f}]")
        for name, prec, folded in (("(a) bf16, LayerNorm then projection (current)", "bf16", False), ("(b) bf16, folded", "bf16", True), ("(c) fp32, folded", "fp32", True)):
            outs = run(make(spec, sd, prec, folded), enc, prompt, ids, steps)
            err = max(float((a - b).abs().max()) for a, b in zip(outs, f32))
            rms = (sum(float(((a - b) ** 2).mean()) for a, b in zip(outs, f32)) / len(outs)) ** 0.5
            agree = sum(int((a[:, :1024].argmax(-1) == b[:, :1024].argmax(-1)).sum()) for a, b in zip(outs, f32)) / (len(outs) * outs[0].shape[0])
            print(f"   {name:48s} max |dlogit| {err:.3e}  rms {rms:.3e}  identical arg-max {100 * agree:5.1f} %")
    print(f"# ({time.time() - t0:.0f} s on the CPU)")


if __name__ == "__main__":
    main()
