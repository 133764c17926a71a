"""CPU: the streamer's halo (streamer.receptive_halo_frames: how many latent frames of left context a chunk decode needs so that
its samples equal the full decode's) checked NUMERICALLY on the oracle codec with the real stride / dilation structure and narrow
channels: changing a code more than `halo` frames to the left of a frame must leave that frame's samples bit-identical, and the
bound must not be grossly loose (a change `halo // 2` frames to the left does reach them for the 44.1 kHz strides)."""
import pytest
import torch

from oracle import dac_oracle as DA
from parler_tts_amd.streamer import receptive_halo_frames


@pytest.mark.parametrize("rates", [(8, 8, 4, 2), (4, 2, 2, 2), (2, 2), (8, 5, 4, 2)])
@torch.no_grad()
def test_halo_covers_the_decoder_receptive_field(rates):
    spec = DA.DacSpec(num_codebooks=3, codebook_size=64, latent_dim=16, decoder_dim=64,
                      decoder_rates=rates, encoder_dim=8)
    sd = DA.make_dac_weights(spec, seed=9)
    orc = DA.DacOracle(spec, sd)
    halo = receptive_halo_frames(rates)
    T = 2 * halo + 12
    t0 = halo + 6
    hop = spec.hop_length
    g = torch.Generator().manual_seed(1)
    codes = torch.randint(0, spec.codebook_size, (1, spec.num_codebooks, T), generator=g)
    wav = orc.decode(codes)[0, 0]

    def changed(frame):
        c = codes.clone()
        c[0, :, frame] = (c[0, :, frame] + 7) % spec.codebook_size
        return orc.decode(c)[0, 0]

    far = changed(t0 - halo - 1)      # one frame beyond the halo: must not reach frame t0 or anything after it
    assert torch.equal(far[t0 * hop:], wav[t0 * hop:])
    assert not torch.equal(far[: t0 * hop], wav[: t0 * hop])
    near = changed(t0 - 1)            # the neighbouring frame does
    assert not torch.equal(near[t0 * hop: (t0 + 1) * hop], wav[t0 * hop: (t0 + 1) * hop])
    # smallest left context that is enough, found by scanning: the formula is an upper bound and within 2x of it
    need = 0
    for h in range(1, halo + 1):
        if not torch.equal(changed(t0 - h)[t0 * hop:], wav[t0 * hop:]):
            need = h
    assert need <= halo and halo <= 2 * need + 2, (need, halo)
    # the same on the right: samples of frames < t0 do not depend on frames >= t0 + halo (the overlapped non-streaming decode keeps one
    # halo of frames back as right context)
    right = changed(t0 + halo)
    assert torch.equal(right[: t0 * hop], wav[: t0 * hop])
