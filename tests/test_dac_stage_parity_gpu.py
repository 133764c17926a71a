"""GPU: every DAC decode kernel pinned against the oracle's restatement of ONE stage on IDENTICAL inputs (ptts_dac_debug_decode_upto).

Why per stage: end to end, two evaluations of the SAME bf16-operand model that differ only in fp32 summation order are already ~1 % apart on
the synthetic 44.1 kHz stack (a 1e-7 perturbation moves a few activations across a bf16 rounding boundary, and ~30 layers of Snake / conv
amplify each 2^-8 flip ~70x; measured on the CPU oracle itself, profiles/r04_dac_bf16_sensitivity.txt). Stage by stage that amplification
is gone: stage s of the engine and of the oracle read the same tensors (the ENGINE's outputs of stage s - 1), so what is left is
  * the fp32 residual stream `raw`: summation order of the MFMA accumulation (1e-6) and, for a residual unit, bf16 flips of its INNER
    activation y (v_sin_f32 / summation order at a rounding boundary): each flip moves one k1-conv operand by 2^-8;
  * the bf16 activation `act` = bf16(Snake(raw)): compared through the ENGINE's raw, so only flips at that one rounding are left.
Bars (bf16-operand mode; set from the first MI355X run x 2, record profiles/r04_parity_dac_stages.txt): relative RMS of the stage's own
contribution to the stream (a unit's raw - raw_in; a transposed conv's raw) <= RAW_TOL (1.5e-4 = 3 x measured; the un-rounded-intermediate negative control is 1.6e-3);
act: every element within one bf16 ulp (2^-7 relative) and <= ACT_FLIP of the elements different at all. The negative controls in
tests/test_oracle_dac.py show what these bars catch: a unit whose inner activation is NOT rounded, a dropped tap, a one-frame halo slip."""
import pytest
import torch

from helpers import log_parity
from oracle import dac_oracle as DA

pytestmark = pytest.mark.gpu

RAW_TOL = {"bf16": 1.5e-4, "fp32": 1e-5}  # measured on MI355X: 5.2e-5 / 1.6e-6 (profiles/r04_parity_dac_stages.txt)
ACT_FLIP = 2e-4                           # measured 2.0e-5


def _rel(a, b):
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-20))


@pytest.mark.parametrize("mode,fuse", [("bf16", True), ("bf16", False), ("bf16", "xin"), ("bf16", "no-xin"), ("fp32", True)])
def test_every_stage_matches_the_oracle_on_identical_inputs(mode, fuse, monkeypatch):
    """fuse = "xin" / "no-xin" (round 6): every fused block / no block on the units that read the fp32 stream (PTTS_DAC_XIN=7 / 0; `True` is the
    default mask). The stop stage of an XIN block also writes its activation (the test reads it); its INPUT path is the production one."""
    from parler_tts_amd.engine import DacEngine

    if fuse in ("xin", "no-xin"):  # (the fused residual units exist in the bf16-operand mode only: fp32 runs once)
        monkeypatch.setenv("PTTS_DAC_XIN", "7" if fuse == "xin" else "0")
    elif not fuse:
        monkeypatch.setenv("PTTS_DAC_NO_FUSE_RES", "1")
    spec = DA.DAC_44KHZ
    sd = DA.make_dac_weights(spec, seed=4321)
    B, T = 2, 37  # 37 latent frames = 296 / 2368 / 9472 / 18944 rows: ragged last tile at every rate, several 128-row tiles from block 1 on
    codes = torch.randint(0, 1024, (B, 9, T), generator=torch.Generator().manual_seed(8))
    d = DacEngine(max_batch=B, max_frames=T, compute_dtype=torch.bfloat16 if mode == "bf16" else torch.float32)
    d.load_state_dict(sd)
    orc = DA.DacOracle(spec, sd, precision=mode)
    z = orc.from_codes(codes)
    prev_act, prev_raw = (DA._rb(z) if mode == "bf16" else z), None
    worst_raw, worst_flip, fails, rows = 0.0, 0.0, [], []

    def check(ok, *what):  # every stage is measured and logged before the first failure is raised
        if not ok:
            fails.append(what)

    for s in range(orc.n_stages()):
        act, raw = d.debug_stage(codes.cuda(), s)
        act, raw = act.cpu(), (raw.cpu() if raw is not None else None)
        exp_raw, _, _ = orc.stage(s, prev_act, prev_raw)
        if raw is not None:
            # a residual unit is judged on what IT adds to the stream (raw - raw_in: the k1 conv's output), not on the stream, whose
            # skip term would dilute an error of the unit's own arithmetic by ~3x; a transposed conv on its output
            unit = s >= 1 and (s - 1) % 4 != 0
            e = _rel(raw - prev_raw, exp_raw - prev_raw) if unit else _rel(raw, exp_raw)
            worst_raw = max(worst_raw, e)
            rows.append(f"s{s}:{e:.1e}")
            check(raw.shape == exp_raw.shape and e <= RAW_TOL[mode], mode, fuse, s, "raw", e)
            check(float((raw - exp_raw).abs().max()) <= 2e-2 * float(exp_raw.abs().max()), mode, fuse, s, "raw max")
        if raw is None:
            # stage 0 hands out no stream: its activation is compared with the oracle's whole stage (conv on the oracle's own latents, Snake, round).
            # The pre-activations differ by fp32 summation order (7168 products), so a few elements land on the other side of a bf16
            # boundary (2^-8 of the element) and near-zero elements differ by the summation noise itself: judged on the relative RMS
            exp_act = orc.stage(s, prev_act, prev_raw)[1]
            e0 = _rel(act, exp_act)
            rows.append(f"s0(act):{e0:.1e}")
            check(e0 <= (2e-3 if mode == "bf16" else 2e-5), mode, fuse, s, "act0", e0)
            worst_raw = max(worst_raw, e0 if mode == "fp32" else 0.0)
        else:
            # act = round(Snake(stream)) with the NEXT layer's alpha, recomputed from the ENGINE's stream: only this one rounding (and v_sin_f32
            # against sin: both evaluate the same fp32 argument) can differ - at most one bf16 ulp on an element, on few elements
            exp_act = _act_of(orc, s, raw)
            diff = (act - exp_act).abs()
            scale = float(exp_act.abs().mean())
            tol = (2.0 ** -7) * exp_act.abs() + 1e-5 * scale if mode == "bf16" else 1e-5 * (scale + exp_act.abs())
            frac = float((diff > 1e-5 * (scale + exp_act.abs())).float().mean())
            worst_flip = max(worst_flip, frac)
            check(bool((diff <= tol).all()), mode, fuse, s, "act ulp", float((diff / tol).max()))
            if mode == "bf16":
                check(frac <= ACT_FLIP, mode, fuse, s, "act flips", frac)
        prev_act, prev_raw = act, raw
    # the last kernel of the decode - Conv1d(C -> 1, k7) + tanh (conv_out_tanh_lds_kernel; fp32 fma chain in both modes) - on the ENGINE's own last
    # activation (it was covered end to end only): what is left is the summation order of 672 fp32 products in front of the tanh
    wave = d.decode(codes.cuda()).cpu().reshape(B, 1, -1)
    exp_wave = orc.final_stage(prev_act)
    e_out = _rel(wave, exp_wave)
    rows.append(f"out:{e_out:.1e}")
    check(wave.shape == exp_wave.shape and e_out <= 1e-5, mode, fuse, "final conv + tanh", e_out)
    check(float((wave - exp_wave).abs().max()) <= 1e-5, mode, fuse, "final conv + tanh max", float((wave - exp_wave).abs().max()))
    log_parity(f"[dac stages {mode} fused={fuse}] {orc.n_stages()} stages + the final conv, {B} x {T} frames: worst raw-stream relative RMS {worst_raw:.2e}, "
               f"worst fraction of activation elements differing (<= 1 bf16 ulp each) {worst_flip:.2e}; per stage: {' '.join(rows)}"
               + (f"; FAILED: {fails[:6]}" if fails else ""), name="r04_parity_dac_stages.txt")
    d.close()
    assert not fails, fails[:6]


def _act_of(orc, s, raw):
    """Snake of stage s's stream with the next layer's alpha, rounded as decode_latents rounds it (DacOracle.stage's second output, from `raw`)."""
    w, d = orc.w, "decoder.model."
    n = len(orc.spec.decoder_rates)
    rb = DA._rb if orc.precision == "bf16" else (lambda v: v)
    if s == 0:
        return rb(DA.snake1d(raw, w[d + "1.block.0.alpha"]))
    bi, k = divmod(s - 1, 4)
    b = f"{d}{bi + 1}.block."
    if k == 0:
        return rb(DA.snake1d(raw, w[b + "2.block.0.alpha"]))
    ri = k - 1
    if ri < 2:
        return rb(DA.snake1d(raw, w[f"{b}{ri + 3}.block.0.alpha"]))
    if bi + 1 < n:
        return rb(DA.snake1d(raw, w[f"{d}{bi + 2}.block.0.alpha"]))
    return DA.snake1d(raw, w[f"{d}{n + 1}.alpha"])
