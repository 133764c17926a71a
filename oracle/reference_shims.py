"""TEST INFRASTRUCTURE ONLY — never imported by the product path (parler_tts_amd/).

Imports the *reference's own* decoder classes (``/root/reference/parler_tts``) inside this
container so that oracle/ restatements can be pinned against them and golden vectors generated.
``/root/reference`` does not exist on the GPU box: only oracle/make_golden.py and the
``-m "not gpu"`` pinning tests (skipped when the tree is absent) may call this.

Three shims, applied from OUTSIDE the reference tree:
  1. a stub ``dac.model.DAC`` module (dac_wrapper/modeling_dac.py:2 imports descript-audio-codec,
     which is not installed and is never executed by the decoder-level oracle);
  2. ``transformers.pytorch_utils.isin_mps_friendly = torch.isin`` (logits_processors.py:2; the
     symbol was removed in transformers 5.x);
  3. read-only ``DynamicCache.key_cache / value_cache`` views over ``layers[i].keys/.values``
     (modeling_parler_tts.py:532-533, :874-875 read ``.key_cache[layer_idx]``).
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("PTTS_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "parler_tts"))


def import_reference():
    """Returns the reference ``parler_tts`` package (imported with bytecode writing disabled)."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True  # root bypasses the read-only bits; never write into /root/reference
    import torch
    import transformers
    import transformers.pytorch_utils as pu
    from transformers.cache_utils import DynamicCache

    if "dac" not in sys.modules:
        dac = types.ModuleType("dac")
        dac_model = types.ModuleType("dac.model")

        class DAC(torch.nn.Module):  # never constructed by the decoder-level oracle
            def __init__(self, *a, **k):
                super().__init__()

        dac_model.DAC = DAC
        dac.model = dac_model
        sys.modules["dac"] = dac
        sys.modules["dac.model"] = dac_model
    if not hasattr(pu, "isin_mps_friendly"):
        pu.isin_mps_friendly = torch.isin
    if not hasattr(DynamicCache, "key_cache"):
        DynamicCache.key_cache = property(lambda self: [l.keys for l in self.layers])
        DynamicCache.value_cache = property(lambda self: [l.values for l in self.layers])
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import parler_tts  # noqa: the reference package

    return parler_tts
