import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from oracle import decoder_oracle as DO
from helpers import make_engine, spec_from_gold, t
from conftest import GOLD
g = np.load(os.path.join(GOLD, "decoder_sin.npz"))
spec = spec_from_gold(g["spec"])
sd = DO.make_decoder_weights(spec, seed=int(g["weight_seed"]))
for dtype, prec in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
    eng = make_engine(spec, sd, dtype)
    orc = DO.DecoderOracle(spec, sd, precision=prec)
    K = spec.num_codebooks; bsz = g["enc"].shape[0]
    ref = orc.forward(torch.full((bsz * K, 1), spec.bos_token_id), t(g["enc"]), t(g["enc_mask"]), t(g["prompt"]), t(g["prompt_mask"]))[:, -1]
    eng.set_gen_params(max_length=16)
    eng.prefill(t(g["enc"]), t(g["enc_mask"]), t(g["prompt"]), t(g["prompt_mask"]), sample=False)
    out = eng.logits().cpu()
    print(prec, "prefill err", float((out - ref).abs().max()), "nan", bool(torch.isnan(out).any()), flush=True)
    for s in range(3):
        ids = t(g["step_ids"][s])
        eng.push_tokens(ids[:, 0]); eng.step_forward()
        r = orc.forward(ids)[:, -1]
        o = eng.logits().cpu()
        print(prec, "step", s, "err", float((o - r).abs().max()), flush=True)
    eng.close()
