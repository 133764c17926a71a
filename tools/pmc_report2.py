"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of tools/prof_eager.py) into HBM bytes per decode
step, taken over the LAST `steps` eager steps of the run (a step = the dispatches from one push_tokens_kernel to the next), so
the figure is at the context the run ends at. gfx950 correction per /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE (KB)
tallies 128-B requests at 64 B -> doubled for these 16 B/lane coalesced streams; WRITE_SIZE uncalibrated (small here).
    python tools/pmc_report2.py fetch.db write.db <steps> <context> <bs> out.json"""
import json, sqlite3, sys
from collections import defaultdict

fetch_db, write_db, steps, context, bs, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]


def per_step(path):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)").fetchall()]
    order = next(c for c in ("dispatch_id", "id", "start", "timestamp") if c in cols)
    rows = cur.execute(f"select {order}, kernel_name, value from counters_collection order by {order}").fetchall()
    marks = [i for i, r in enumerate(rows) if "push_tokens_kernel" in r[1]]
    lo, hi = marks[-steps - 1], marks[-1]  # the last `steps` complete periods
    by = defaultdict(lambda: [0.0, 0])
    tot = 0.0
    for _, nm, v in rows[lo:hi]:
        key = nm.split("(")[0].replace("void ", "")[:70]
        by[key][0] += v
        by[key][1] += 1
        tot += v
    return tot / steps, {k: (v[0] / steps, v[1] / steps) for k, v in by.items()}


f_tot, f_by = per_step(fetch_db)
w_tot, w_by = per_step(write_db)
print(f"{'kernel (eager decode step, bs=' + str(bs) + ')':72s} {'launches/step':>13s} {'fetch MB/step (x2)':>19s} {'write MB/step':>14s}")
for k in sorted(f_by, key=lambda k: -f_by[k][0]):
    print(f"{k:72s} {f_by[k][1]:13.1f} {f_by[k][0] * 2 * 1024 / 1e6:19.3f} {w_by.get(k, (0, 0))[0] * 1024 / 1e6:14.3f}")
traffic = f_tot * 2 * 1024 + w_tot * 1024
L, H, F, K, V, N = 24, 1024, 4096, 9, 1088, 64
alg = (L * (6 * H * H + 2 * H * F) + K * V * H) * 2 + bs * 2 * L * H * (context + N) * 2 + bs * (K * H * 2 + K * V * 4)
print(f"HBM traffic per decode step (Mini-v1 bf16 bs={bs}, self-KV context ~{context}): {traffic / 1e6:.1f} MB; algorithmic {alg / 1e6:.1f} MB -> ratio {traffic / alg:.3f}")
json.dump({"traffic_bytes_per_step": traffic, "context": context, "bs": bs, "dtype": "bf16", "algorithmic_mb": round(alg / 1e6, 1),
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/prof_eager.py (eager launches of the step's kernels), FETCH x2 (gfx950)"},
          open(out, "w"))
