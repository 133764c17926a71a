"""CPU property test (hypothesis): the product's closed-form delay pattern (modeling_parler_tts.build_delay_pattern_mask /
apply_delay_pattern_mask) against the oracle restatement of the reference's loop-built one (modeling:205-276; pinned against the
reference class itself in test_oracle_golden.py / test_oracle_vs_reference.py, which also runs this property against the reference
function directly where the tree is mounted) over random codebook counts, batch sizes, given-column counts and max lengths, including the
degenerate max_length < 2K - 1 case (:241-243) where no delay is applied."""
import pytest
import torch

hypothesis = pytest.importorskip("hypothesis")
from hypothesis import given, settings, strategies as st  # noqa: E402

import parler_tts_amd as P  # noqa: E402
from oracle import decoder_oracle as DO  # noqa: E402


@settings(max_examples=150, deadline=None)
@given(K=st.integers(1, 9), bsz=st.integers(1, 3), seq_len=st.integers(1, 12), extra=st.integers(0, 30), seed=st.integers(0, 10_000))
def test_closed_form_equals_restated_and_reference_pattern(K, bsz, seq_len, extra, seed):
    max_len = seq_len + extra
    # the reference's loop writes codebook k's given columns at [k, seq_len + k): it only runs (max_len >= 2K - 1, :241-243) when they fit
    hypothesis.assume(max_len < 2 * K - 1 or max_len >= seq_len + K - 1)
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, 1024, (bsz * K, seq_len), generator=g)
    ids[:, 0] = 1025
    want_ids, want_mask = DO.build_delay_pattern_mask(ids, 1025, 1024, max_len, K)
    got_ids, got_mask = P.build_delay_pattern_mask(ids, 1025, 1024, max_len, K)
    assert torch.equal(got_ids, want_ids) and torch.equal(got_mask, want_mask)
    # applying the mask to a longer generated sequence: forced positions come from the mask, free ones (-1) from the sequence
    n = min(max_len, got_ids.shape[1] + int(torch.randint(0, 5, (1,), generator=g)))
    seq = torch.randint(0, 1024, (bsz * K, n), generator=g)
    a = P.apply_delay_pattern_mask(seq, got_mask)
    b = DO.apply_delay_pattern_mask(seq, want_mask)
    assert torch.equal(a, b)
    m = got_mask[:, :n]
    assert torch.equal(a, torch.where(m == -1, seq, m))
