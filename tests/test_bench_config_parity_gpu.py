"""GPU parity ON THE CONFIGURATION bench.py MEASURES (BASELINE.json configs[1] / configs[2]): full parler-tts-mini-v1
decoder (24 layers, H = 1024, F = 4096, K = 9, V = 1088), 64 description + 32 prompt tokens, 860 frames = 868 decoder
passes, the model / seeds / inputs built by bench.py itself (build_model, synthetic_batch).

  (a) fp32, bs = 1 : teacher-forced logits vs the oracle at EVERY one of the 868 passes (<= 2e-4; measured ~1e-5), then the
      free-running graph path: ids bit-exact up to the first pass whose oracle top-2 margin is < 2e-4 (column reported).
  (b) bf16, bs = 1 and bs = 32 : 64 teacher-forced passes vs the bf16-quantised oracle (same bf16 weights / KV / Linear
      inputs, fp32 accumulate): max |dlogit| <= TOL_BF16 and every arg-max disagreement must sit on an oracle margin
      smaller than twice the measured error (i.e. only near-ties may flip); agreement fraction >= 98 %.
  (c) DAC: 860 frames through the 44.1 kHz stack, exact-f32 mode RMS <= 1e-4, bf16-operand mode <= 3 % of the signal RMS.

TOL_BF16 is set from measurement, not by fiat: the engine and the oracle evaluate the SAME quantised model and differ by
summation order and by where an fp32 value lands relative to a bf16 rounding boundary (a 1-ulp flip of a Linear input is
2^-8 relative); over 24 layers the measured max |dlogit| at these shapes is recorded in profiles/r02_parity_bench_config.txt.
"""
import os
import sys
import time

import pytest
import torch

from oracle import dac_oracle as DA
from oracle import decoder_oracle as DO

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

TOL_FP32 = 2e-4
TOL_BF16 = 2.5e-2   # logits are O(0.3-1.0); measured max error: see profiles/r02_parity_bench_config.txt
LOG = os.path.join(ROOT, "gpurun_out", "r02_parity_bench_config.txt")


def _log(msg):
    print(msg, flush=True)
    try:
        os.makedirs(os.path.dirname(LOG), exist_ok=True)
        with open(LOG, "a") as f:
            f.write(msg + "\n")
    except OSError:
        pass


_MASTER = []  # bench.build_model's fp32 CPU model, built once per session (its random init of ~1.2 G parameters takes ~25 s of box time)


def _bench_model(dtype, dev=None):
    """The model bench.py benchmarks: `bench.build_model` constructs it in fp32 on the CPU and casts / moves it; here the CPU master is
    kept and every test gets its own copy cast and moved the same way (identical tensors, checked on the CPU in tests/test_bench_line_cpu.py)."""
    import copy

    import bench

    dev = dev if dev is not None else torch.device("cuda", 0)
    if not _MASTER:
        _MASTER.append(bench.build_model(0, 1, torch.device("cpu"), torch.float32))
    model = copy.deepcopy(_MASTER[0]).to(device=dev, dtype=dtype)
    return bench, model, dev


def _conditioning(bench, model, bs, dev):
    desc, prompt = bench.synthetic_batch(bs, 0, dev)
    enc = model._encode_description(desc, None).float()
    pr = model.embed_prompts(prompt).float()
    return enc, pr


def _decoder_sd(model):
    return {k: v.detach().float().cpu() for k, v in model.decoder.state_dict().items()}


def _teacher_forced_engine(eng, enc, pr, seq, n_pass):
    """logits of passes 0..n_pass-1 when the engine is fed the oracle's own columns (the eager launch path: the same kernels
    the captured graph holds, launched one by one)."""
    eng.prefill(enc, None, pr, None, sample=False)
    out = [eng.logits().cpu()]
    for s in range(1, n_pass):
        eng.push_tokens(seq[:, s])
        eng.step_forward()
        out.append(eng.logits().cpu())
    return out


def _first_unsafe(step_logits, spec, L):
    """number of columns determined by passes whose oracle top-2 margin is >= 2e-4 (the fp32 noise band is ~1e-5)"""
    for s_, lg in enumerate(step_logits):
        lg = lg.clone()
        lg[:, spec.eos_token_id] = -float("inf")  # min_new_tokens blocks EOS on every pass of this run
        top2 = torch.topk(lg, 2, dim=-1)[0]
        if float((top2[:, 0] - top2[:, 1]).min()) < 2e-4:
            return s_ + 1  # columns [0, s_] are determined by safe passes
    return L


def _oracle_run_from_golden_ids(spec, sd, enc, pr, L):
    """The oracle's 868-pass greedy run without 868 sequential CPU passes on the GPU box: the ids of that run are a committed fixture
    (tests/golden/bench_parity_ids.npz, oracle/make_bench_parity_golden.py); ONE batched causal forward teacher-forced on them gives
    the logits of every pass (same arithmetic as the cached steps up to summation order, ~1e-6). The fixture is only an accelerator:
    unless it is this machine's oracle's own arg-max on every pass before the first unsafe margin, None is returned and the caller
    runs the sequential loop."""
    import numpy as np

    path = os.path.join(ROOT, "tests", "golden", "bench_parity_ids.npz")
    if os.environ.get("PTTS_PARITY_SEQUENTIAL") or not os.path.exists(path):
        return None
    ids = torch.from_numpy(np.load(path)["ids"].astype(np.int64))
    if tuple(ids.shape) != (spec.num_codebooks, L):
        return None
    _, pattern = DO.build_delay_pattern_mask(ids[:, :1], spec.bos_token_id, spec.pad_token_id, L, spec.num_codebooks)
    fed = DO.apply_delay_pattern_mask(ids, pattern)[:, : L - 1]
    with torch.no_grad():
        lg = DO.DecoderOracle(spec, sd).forward(fed, enc, None, pr, None)[:, -(L - 1):]  # [rows, pass, V]: pass s predicts column s + 1
    step_logits = [lg[:, s].contiguous() for s in range(L - 1)]
    safe = _first_unsafe(step_logits, spec, L)
    m = lg[:, : safe - 1].clone()
    m[..., spec.eos_token_id] = -float("inf")
    if not torch.equal(m.argmax(-1), ids[:, 1:safe]):
        return None
    return ids, step_logits


def test_fp32_bs1_all_868_passes_and_free_running_ids():
    bench, model, dev = _bench_model(torch.float32)
    spec = DO.MINI_V1
    enc, pr = _conditioning(bench, model, 1, dev)
    sd = _decoder_sd(model)
    L = bench.NEW_TOKENS + 1
    gp = DO.GenParams(max_length=L, min_new_tokens=bench.NEW_TOKENS)
    torch.set_num_threads(min(os.cpu_count() or 8, 16))
    t0 = time.time()
    try:
        fast = _oracle_run_from_golden_ids(spec, sd, enc.cpu(), pr.cpu(), L)
    except Exception as e:  # noqa: BLE001 — the fixture is an accelerator only: any problem with it means the sequential loop, not a failure
        _log(f"[fp32 bs=1] golden-ids fast path unavailable ({e!r}): running the sequential oracle loop")
        fast = None
    if fast is not None:
        import types

        ref = types.SimpleNamespace(sequences=fast[0], step_logits=fast[1])
    else:
        ref = DO.sample_loop(DO.DecoderOracle(spec, sd), enc.cpu(), None, pr.cpu(), None, gp, keep_logits=True)
    t_or = time.time() - t0
    assert ref.sequences.shape[1] == L and len(ref.step_logits) == bench.NEW_TOKENS
    safe = _first_unsafe(ref.step_logits, spec, L)  # first pass whose arg-max margin is inside the fp32 noise band
    eng = model._get_engine(1, bench.N_DESC, bench.N_PROMPT, L)
    eng.set_gen_params(max_length=L, min_new_tokens=bench.NEW_TOKENS)
    outs = _teacher_forced_engine(eng, enc, pr, ref.sequences.to(dev), bench.NEW_TOKENS)
    errs = torch.tensor([float((a - b).abs().max()) for a, b in zip(outs, ref.step_logits)])
    worst, at = float(errs.max()), int(errs.argmax())
    _log(f"[fp32 bs=1] 868 teacher-forced passes: max |dlogit| {worst:.2e} at pass {at} (mean of per-pass max {float(errs.mean()):.2e}); "
         f"oracle {t_or:.0f} s ({'one batched forward on the golden ids' if fast is not None else 'sequential free run'}); "
         f"first pass with oracle margin < 2e-4: {safe} of {L} columns")
    assert worst <= TOL_FP32, (worst, at)
    assert safe >= 8, f"oracle margins too small too early ({safe})"
    ids = eng.generate_ids(enc, None, pr, None).cpu()  # free-running: prefill + 867 hipGraph replays
    assert ids.shape == ref.sequences.shape
    assert torch.equal(ids[:, :safe], ref.sequences[:, :safe]), "greedy ids differ before the first unsafe pass"
    agree = float((ids == ref.sequences).float().mean())
    _log(f"[fp32 bs=1] free-running graph path: ids bit-exact on columns [0, {safe}); overall agreement with the oracle run {agree * 100:.1f} %")


@pytest.mark.parametrize("bs", [1, 32])
def test_bf16_logits_and_argmax_vs_quantised_oracle(bs):
    bench, model, dev = _bench_model(torch.bfloat16)
    spec = DO.MINI_V1
    enc, pr = _conditioning(bench, model, bs, dev)
    sd = _decoder_sd(model)
    n_pass = 65
    L = bench.NEW_TOKENS + 1
    torch.set_num_threads(min(os.cpu_count() or 8, 16))
    orc = DO.DecoderOracle(spec, sd, precision="bf16")
    gp = DO.GenParams(max_length=n_pass + 1, min_new_tokens=n_pass)
    ref = DO.sample_loop(orc, enc.cpu(), None, pr.cpu(), None, gp, keep_logits=True)  # the oracle's own greedy run: the columns to feed
    eng = model._get_engine(bs, bench.N_DESC, bench.N_PROMPT, L)
    eng.set_gen_params(max_length=n_pass + 1, min_new_tokens=n_pass)
    outs = _teacher_forced_engine(eng, enc, pr, ref.sequences.to(dev), n_pass)
    worst, n_rows, n_same, unexplained = 0.0, 0, 0, 0
    for a, b in zip(outs, ref.step_logits):
        a = a.clone(); b = b.clone()
        err = float((a - b).abs().max())
        worst = max(worst, err)
        a[:, spec.eos_token_id] = -float("inf"); b[:, spec.eos_token_id] = -float("inf")
        ia, ib = a.argmax(-1), b.argmax(-1)
        top2 = torch.topk(b, 2, dim=-1)[0]
        margin = top2[:, 0] - top2[:, 1]
        diff = ia != ib
        n_rows += ia.numel()
        n_same += int((~diff).sum())
        unexplained += int((diff & (margin > 2 * err)).sum())  # a flip is legitimate only inside the error band
    frac = n_same / n_rows
    _log(f"[bf16 bs={bs}] {n_pass} teacher-forced passes x {bs * 9} rows: max |dlogit| {worst:.2e}; identical arg-max {frac * 100:.2f} % "
         f"({n_rows - n_same} flips, {unexplained} outside the 2x error band)")
    assert worst <= TOL_BF16, worst
    assert unexplained == 0
    assert frac >= 0.98, frac


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_dac_860_frames_vs_oracle(mode):
    from parler_tts_amd.engine import DacEngine
    from parler_tts_amd.synthetic import random_dac_state_dict

    T = 860
    dsd = random_dac_state_dict(seed=4321)
    codes = torch.randint(0, 1024, (1, 9, T), generator=torch.Generator().manual_seed(7))
    torch.set_num_threads(min(os.cpu_count() or 8, 16))
    t0 = time.time()
    ref = DA.DacOracle(DA.DAC_44KHZ, dsd).decode(codes)[0, 0]
    t_or = time.time() - t0
    dac = DacEngine(max_batch=1, max_frames=T, compute_dtype=torch.float32 if mode == "f32" else torch.bfloat16)
    dac.load_state_dict({k: v.cuda() for k, v in dsd.items()})
    wav = dac.decode(codes.cuda())[0, 0].cpu()
    assert wav.shape == ref.shape == (T * 512,)
    rms_err = float((wav - ref).pow(2).mean().sqrt())
    rms_sig = float(ref.pow(2).mean().sqrt())
    _log(f"[dac {mode}] {T} frames: waveform RMS error {rms_err:.2e} (signal RMS {rms_sig:.2e}, ratio {rms_err / rms_sig:.2e}); oracle {t_or:.0f} s")
    if mode == "f32":
        assert rms_err <= 1e-4, rms_err
    else:
        assert rms_err <= 0.03 * rms_sig, (rms_err, rms_sig)
