"""Generates tests/golden/t5_tiny.npz: outputs of the INSTALLED transformers T5EncoderModel (the class the reference instantiates as its text
encoder, /root/reference/parler_tts/modeling_parler_tts.py:2345-2348) on seeded synthetic weights, for the HIP description encoder's parity
tests on the GPU box. Run from the repo root: python oracle/make_golden_t5.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import t5_oracle as TO  # noqa: E402


def main():
    spec = TO.T5Spec(vocab_size=160, d_model=128, d_kv=64, d_ff=256, num_layers=2, num_heads=2)
    seed = 21
    sd = TO.make_t5_weights(spec, seed=seed)
    g = torch.Generator().manual_seed(22)
    ids = torch.randint(0, spec.vocab_size, (3, 37), generator=g)
    mask = torch.ones(3, 37, dtype=torch.long)
    mask[0, 30:] = 0
    mask[2, :9] = 0
    hf = TO.hf_encoder(spec, sd)
    with torch.no_grad():
        masked = hf(input_ids=ids, attention_mask=mask).last_hidden_state
        unmasked = hf(input_ids=ids).last_hidden_state
    keys = ["vocab_size", "d_model", "d_kv", "d_ff", "num_layers", "num_heads", "relative_attention_num_buckets", "relative_attention_max_distance",
            "layer_norm_epsilon"]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "t5_tiny.npz"), spec_keys=np.array(keys), spec_vals=np.array([float(getattr(spec, k)) for k in keys]),
                        seed=np.int64(seed), ids=ids.numpy(), mask=mask.numpy(), hf_masked=masked.numpy(), hf_unmasked=unmasked.numpy())
    print("wrote tests/golden/t5_tiny.npz", tuple(masked.shape))


if __name__ == "__main__":
    main()
