"""GPU tests of the parked experiments (NOT part of the suite the driver runs: `pytest tests/` does not collect this directory).
Run against a variant library built by tools/experimental/build_variants.sh:

    PTTS_LIB=parler_tts_amd/exp/libptts_lm_batch32_both.so python -m pytest tools/experimental/test_experimental_gpu.py -q -k "arriver or groups"
    PTTS_LIB=parler_tts_amd/exp/libptts_dac_fused_resunit.so python -m pytest tools/experimental/test_experimental_gpu.py -q -k fused_residual

Each test switches its path on through the environment variable the variant library reads (at engine creation for the LM flags, per call
for the DAC flag) and compares with the oracle at the tolerances of the default path. With the default library the flags are ignored and
the tests exercise the default kernels."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import dac_oracle as DA  # noqa: E402
from oracle import decoder_oracle as DO  # noqa: E402
from test_lm_gpu import _teacher_forced_vs_oracle  # noqa: E402
from test_dac_gpu import _rel_rms  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bsz", [12, 32])
def test_fc2_last_arriver_combine(bsz, monkeypatch):
    """PTTS_FC2_ARRIVE=1: the last split-K workgroup of every fc2 strip folds the partials + residual in split order and writes the
    LayerNorm strip statistics, so LN1 of the next layer runs PRO_LNS too (no rows_prep node). 3 layers: two arrive rounds per forward,
    counters re-armed between launches; 4 steps; the default path's tolerances."""
    monkeypatch.setenv("PTTS_FC2_ARRIVE", "1")
    spec = DO.DecoderSpec(num_hidden_layers=3, max_position_embeddings=512)
    sd = DO.make_decoder_weights(spec, seed=48)
    for dtype, prec, tol in ((torch.float32, "fp32", 5e-5), (torch.bfloat16, "bf16", 2e-2)):
        err = _teacher_forced_vs_oracle(spec, sd, dtype, prec, bsz=bsz, N=21, P=6, steps=4, masks=True, seed=bsz)
        assert err < tol, (bsz, prec, err)


@pytest.mark.parametrize("bsz", [9, 12, 32])
def test_fused_cross_block_in_groups_of_eight(bsz, monkeypatch):
    """PTTS_XATTN_GROUPS=1: at batch 9..32 the cross block's LN2 + q projection + cross-attention run in the batch <= 8 fused kernel, one
    workgroup per (head, group of 8 utterances); ragged last group at 9 and 12."""
    monkeypatch.setenv("PTTS_XATTN_GROUPS", "1")
    spec = DO.DecoderSpec(num_hidden_layers=2, max_position_embeddings=512)
    sd = DO.make_decoder_weights(spec, seed=49)
    for dtype, prec, tol in ((torch.float32, "fp32", 5e-5), (torch.bfloat16, "bf16", 2e-2)):
        err = _teacher_forced_vs_oracle(spec, sd, dtype, prec, bsz=bsz, N=21, P=6, steps=4, masks=True, seed=bsz)
        assert err < tol, (bsz, prec, err)


def test_both_lm_experiments_together(monkeypatch):
    monkeypatch.setenv("PTTS_FC2_ARRIVE", "1")
    monkeypatch.setenv("PTTS_XATTN_GROUPS", "1")
    spec = DO.DecoderSpec(num_hidden_layers=3, max_position_embeddings=512)
    sd = DO.make_decoder_weights(spec, seed=50)
    for dtype, prec, tol in ((torch.float32, "fp32", 5e-5), (torch.bfloat16, "bf16", 2e-2)):
        err = _teacher_forced_vs_oracle(spec, sd, dtype, prec, bsz=32, N=21, P=6, steps=4, masks=True, seed=5)
        assert err < tol, (prec, err)


def test_fused_residual_units_44khz(monkeypatch):
    """PTTS_DAC_FUSE_RES=1 (read per call): the residual units of the two narrow blocks (C = 192, 96) run as one launch each (k7 -> Snake ->
    bf16 tile in LDS -> k1 -> + skip -> Snake). Same arithmetic as the two-launch path up to the rounding of the intermediate to bf16
    (both do it). T = 150 frames spans several 128-frame tiles with a ragged last one at every rate; batch 2."""
    from parler_tts_amd.engine import DacEngine

    spec = DA.DAC_44KHZ
    sd = DA.make_dac_weights(spec, seed=4321)
    codes = torch.randint(0, 1024, (2, 9, 150), generator=torch.Generator().manual_seed(8))
    ref = DA.DacOracle(spec, sd).decode(codes)
    d = DacEngine(max_batch=2, max_frames=160, compute_dtype=torch.bfloat16)
    d.load_state_dict(sd)
    plain = d.decode(codes.cuda()).cpu()
    monkeypatch.setenv("PTTS_DAC_FUSE_RES", "1")
    fused = d.decode(codes.cuda()).cpu()
    monkeypatch.delenv("PTTS_DAC_FUSE_RES")
    for b in range(2):
        assert _rel_rms(fused[b], ref[b]) <= 3e-2, b
        assert _rel_rms(fused[b], plain[b]) <= 1e-2, b


@pytest.mark.parametrize("bsz", [40, 64, 128])
def test_decode_batch_above_32_default_library(bsz):
    """Not an experiment of a patch but of a configuration: decode at batch > 32 per GPU (whole-node throughput lever: the step is
    latency-bound at 32, so utterances per step are nearly free until the KV traffic dominates). The engine takes the prefill-sized code
    path there (rows_prep + 128-row PRO_COPY passes, plain fc2, no producer statistics); the suite only covers batch <= 32. Mini width,
    2 layers, ragged masks, 3 teacher-forced steps vs the oracle at the default tolerances."""
    spec = DO.DecoderSpec(num_hidden_layers=2, max_position_embeddings=512)
    sd = DO.make_decoder_weights(spec, seed=61)
    for dtype, prec, tol in ((torch.float32, "fp32", 5e-5), (torch.bfloat16, "bf16", 2e-2)):
        err = _teacher_forced_vs_oracle(spec, sd, dtype, prec, bsz=bsz, N=21, P=6, steps=3, masks=True, seed=bsz)
        assert err < tol, (bsz, prec, err)
