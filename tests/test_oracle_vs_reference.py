"""CPU, only where the reference tree is mounted (this build container; skipped on the GPU box, which has no
/root/reference): the oracle is re-checked LIVE against the reference's own classes - the same comparisons
oracle/make_golden.py runs before it freezes tests/golden/*.npz. Nothing is written (bytecode writing is disabled by the
import shim, golden files are not touched)."""
import pytest
import torch

from oracle.reference_shims import reference_available

pytestmark = pytest.mark.skipif(not reference_available(), reason="/root/reference is not mounted here")


@pytest.fixture(scope="module")
def ref():
    from oracle.reference_shims import import_reference

    return import_reference()


@pytest.mark.parametrize("variant", ["sin", "rope", "gqa"])
def test_decoder_forward_matches_reference_live(ref, variant):
    import oracle.make_golden as mg

    torch.manual_seed(0)
    worst = mg.gen_decoder(ref, variant, save=False)  # ParlerTTSForCausalLM: prefill with padded masks + 6 cached steps, SDPA and eager
    assert worst < 2e-6


def test_delay_pattern_and_eos_gate_match_reference_live(ref):
    from oracle import decoder_oracle as DO
    import parler_tts_amd as P

    M = ref.modeling_parler_tts
    for ci, (K, seq_len, max_len, bsz) in enumerate([(9, 1, 30, 2), (4, 3, 8, 1), (9, 13, 40, 1), (9, 1, 12, 1), (2, 1, 3, 2)]):
        g = torch.Generator().manual_seed(ci)
        ids = torch.randint(0, 1024, (bsz * K, seq_len), generator=g)
        ids[:, 0] = 1025
        a = M.build_delay_pattern_mask(ids, 1025, 1024, max_len, K)
        for impl in (DO.build_delay_pattern_mask, P.build_delay_pattern_mask):  # oracle restatement and the product's closed form
            b = impl(ids, 1025, 1024, max_len, K)
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), (ci, impl.__module__)
    K, bsz, V = 9, 2, 1088
    proc = M.ParlerTTSLogitsProcessor(1024, K, bsz, "cpu")
    gate, mine = DO.EosGate(1024, K, bsz), P.ParlerTTSLogitsProcessor(1024, K, bsz, "cpu")
    g = torch.Generator().manual_seed(3)
    seq = torch.full((bsz * K, 1), 1025, dtype=torch.long)
    for step in range(20):
        nxt = torch.randint(0, 1024, (bsz * K,), generator=g)
        nxt[torch.rand(bsz * K, generator=g) < 0.15] = 1024  # sprinkle EOS
        seq = torch.cat([seq, nxt[:, None]], dim=1)
        r = proc(seq, torch.zeros(bsz * K, V))
        assert torch.equal(r, gate(seq, torch.zeros(bsz * K, V))) and torch.equal(r, mine(seq, torch.zeros(bsz * K, V))), step
