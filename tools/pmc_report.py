"""Summarise the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, tools/prof_eager.py) into per-kernel
HBM bytes per dispatch and per decode step. gfx950 correction per /opt/skills/guides/MI355X_MICROARCH.md §HBM:
FETCH_SIZE (KB) counts 128-B requests at 64 B -> doubled for these 16 B/lane coalesced streams; WRITE_SIZE uncalibrated."""
import json, re, sqlite3, sys
fetch_db, write_db, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
def load(path):
    cur = sqlite3.connect(path).cursor()
    return cur.execute("select kernel_name, grid_size_x, sum(value), count(*) from counters_collection group by kernel_name, grid_size_x").fetchall()
# decode-step kernel classes at bs=1 (name fragment, grid_x, launches per step); prefill variants (MTP=2 / prep) are excluded
STEP = (("gemm_strip_kernel<unsigned short, 1, 1, 1, true>", 65536, 24, "LN3+fc1+GELU"), ("gemm_strip_kernel<unsigned short, 0, 2, 1, true>", 65536, 24, "fc2+residual"),
        ("gemm_strip_kernel<unsigned short, 1, 0, 1, true>", 49152, 24, "LN1+QKV"), ("xattn_fused_kernel", 8192, 24, "LN2+crossQ+cross-attention"),
        ("gemm_strip_kernel<unsigned short, 2, 2, 1, true>", 16384, 24, "self combine+out_proj+residual"),
        ("gemm_strip_kernel<unsigned short, 3, 2, 1, true>", 16384, 24, "cross out_proj+residual"), ("attn_kernel", 1024, 24, "self-attention (4 splits)"),
        ("gemm_strip_kernel<unsigned short, 1, 0, 1, true>", 156672, 1, "final LN + 9 LM heads"), ("embed_kernel", 256, 1, "embed (eager path only)"),
        ("tail_kernel", 576, 1, "sampler tail"))
rows = {}
for nm, gx, val, n in load(fetch_db):
    rows[(nm, gx)] = [val * 2 * 1024, 0.0, n]
for nm, gx, val, n in load(write_db):
    rows.setdefault((nm, gx), [0.0, 0.0, n])[1] = val * 1024
tot = 0.0
print(f"{'decode-step kernel':34s} {'launches/step':>14s} {'fetch MB/launch (x2 corr.)':>28s} {'write MB/launch':>16s}")
for frag, gx, per_step, label in STEP:
    hit = [(k, v) for k, v in rows.items() if frag in k[0] and k[1] == gx]
    if not hit:
        continue
    f = sum(v[0] for _, v in hit); w = sum(v[1] for _, v in hit); n = sum(v[2] for _, v in hit)
    tot += (f + w) / n * per_step
    print(f"{label:34s} {per_step:14d} {f / n / 1e6:28.3f} {w / n / 1e6:16.3f}")
print(f"HBM traffic per decode step (bs=1 bf16, self-KV context ~40): {tot / 1e6:.1f} MB   (algorithmic: 724.7 MB weights + ~10 MB KV + logits = ~735 MB)")
json.dump({"traffic_bytes_per_step": tot, "context": 40, "bs": 1, "dtype": "bf16", "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/prof_eager.py, FETCH x2 (gfx950)"},
          open(sys.argv[4], "w"))
