"""GPU parity tests of the DAC decode engine (codes → waveform) against the oracle restatement.
Bar (BASELINE.json north_star): waveform RMS error <= 1e-4 in fp32; we also bound the max abs error."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLD
from helpers import make_dac, t
from oracle import dac_oracle as DA

pytestmark = pytest.mark.gpu


def _rms(a, b):
    return float((a - b).pow(2).mean().sqrt())


def test_tiny_matches_golden_waveform():
    g = np.load(os.path.join(GOLD, "dac_tiny.npz"))
    sd = DA.make_dac_weights(DA.DAC_TINY, seed=int(g["weight_seed"]), weight_norm_format="parametrized")
    dac = make_dac(DA.DAC_TINY, sd, max_batch=2, max_frames=32)
    wav = dac.decode(t(g["codes"]).cuda()).cpu()
    ref = t(g["wav"])
    assert wav.shape == ref.shape
    assert _rms(wav, ref) <= 1e-5 and (wav - ref).abs().max() < 1e-4


@pytest.mark.parametrize("fmt", ["folded", "legacy", "parametrized"])
def test_weight_norm_formats(fmt):
    spec = DA.DAC_TINY
    sd = DA.make_dac_weights(spec, seed=99, weight_norm_format=fmt)
    codes = torch.randint(0, 1024, (1, 9, 7), generator=torch.Generator().manual_seed(1))
    ref = DA.DacOracle(spec, sd).decode(codes)
    wav = make_dac(spec, sd, max_batch=1, max_frames=16).decode(codes.cuda()).cpu()
    assert _rms(wav, ref) <= 1e-5


@pytest.mark.parametrize("T", [1, 5, 33, 64])
def test_ragged_lengths_and_edges(T):
    """T=1 (all taps hit the zero padding), T not a multiple of the 32-frame tile, T = tile boundary."""
    spec = DA.DAC_TINY
    sd = DA.make_dac_weights(spec, seed=5)
    codes = torch.randint(0, 1024, (2, 9, T), generator=torch.Generator().manual_seed(T))
    ref = DA.DacOracle(spec, sd).decode(codes)
    wav = make_dac(spec, sd, max_batch=2, max_frames=64).decode(codes.cuda()).cpu()
    assert wav.shape == (2, 1, spec.hop_length * T)
    assert _rms(wav, ref) <= 1e-5


def test_full_size_44khz_stack():
    """The real 44 kHz decoder (1024→1536, strides 8,8,4,2; 54 M parameters) on 24 frames."""
    spec = DA.DAC_44KHZ
    sd = DA.make_dac_weights(spec, seed=4321)
    codes = torch.randint(0, 1024, (1, 9, 24), generator=torch.Generator().manual_seed(2))
    ref = DA.DacOracle(spec, sd).decode(codes)
    wav = make_dac(spec, sd, max_batch=1, max_frames=32).decode(codes.cuda()).cpu()
    assert wav.shape == (1, 1, 512 * 24)
    rms_sig = float(ref.pow(2).mean().sqrt())
    assert rms_sig > 0.05  # the synthetic decoder produces a non-degenerate waveform
    assert _rms(wav, ref) <= 1e-4, _rms(wav, ref)


def test_shift_equivariance_away_from_edges():
    """Size-independent property: away from the edges the decoder is shift-equivariant — decoding codes shifted by
    s frames gives the waveform shifted by s*hop. The tiny spec's small strides (4,2,2,2) give a receptive field of
    about 23 latent frames per side (dilation-9 k7 convs at 4x resolution), so compare >= 28 frames from any edge."""
    spec = DA.DAC_TINY
    sd = DA.make_dac_weights(spec, seed=8)
    g = torch.Generator().manual_seed(3)
    codes = torch.randint(0, 1024, (1, 9, 96), generator=g)
    dac = make_dac(spec, sd, max_batch=1, max_frames=96)
    a = dac.decode(codes[:, :, :80].cuda()).cpu()     # global frames 0..79
    b = dac.decode(codes[:, :, 16:96].cuda()).cpu()   # global frames 16..95
    hop = spec.hop_length
    lo, hi = 44 * hop, 52 * hop
    assert (a[..., lo:hi] - b[..., lo - 16 * hop: hi - 16 * hop]).abs().max() < 1e-5
    assert (a[..., :hop] - b[..., :hop]).abs().max() > 1e-3  # sanity: different content where frames differ


def test_errors():
    spec = DA.DAC_TINY
    sd = DA.make_dac_weights(spec, seed=5)
    dac = make_dac(spec, sd, max_batch=1, max_frames=8)
    with pytest.raises(ValueError, match="max_frames"):
        dac.decode(torch.zeros(1, 9, 9, dtype=torch.long).cuda())
    with pytest.raises(ValueError, match="max_batch"):
        dac.decode(torch.zeros(2, 9, 4, dtype=torch.long).cuda())
    with pytest.raises(ValueError, match="audio_codes"):
        dac.decode(torch.zeros(1, 8, 4, dtype=torch.long).cuda())
