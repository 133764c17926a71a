"""Per-kernel HBM bytes of ONE DAC decode out of two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of `tools/cabi_probe dac <B> reps=1`)
beside the kernel durations of a third --kernel-trace pass: which codec kernels are bandwidth-bound? gfx950 correction as in pmc_report2.py
(FETCH_SIZE in KB tallies 128-byte requests at 64 B: doubled).   python tools/pmc_dac_report.py fetch.db write.db trace.db <decodes in the run>"""
import sqlite3, sys
from collections import defaultdict

fetch_db, write_db, trace_db, ndec = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])


def short(nm):
    nm = (nm or "").replace("void ", "").replace("(anonymous namespace)::", "")
    return nm.split("(")[0][:64]


def counters(path):
    cur = sqlite3.connect(path).cursor()
    by = defaultdict(lambda: [0.0, 0])
    for nm, v in cur.execute("select kernel_name, value from counters_collection").fetchall():
        by[short(nm)][0] += v
        by[short(nm)][1] += 1
    return by


def durations(path):
    cur = sqlite3.connect(path).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
    t = next((x for x in tabs if x == "kernels"), None) or next(x for x in tabs if "kernel" in x.lower() and "dispatch" in x.lower())
    cols = [r[1] for r in cur.execute(f"pragma table_info({t})").fetchall()]
    name = next(c for c in ("name", "kernel_name") if c in cols)
    by = defaultdict(lambda: [0.0, 0])
    for nm, s, e in cur.execute(f"select {name}, start, end from {t}").fetchall():
        by[short(nm)][0] += (e - s) / 1e3
        by[short(nm)][1] += 1
    return by


f, w, d = counters(fetch_db), counters(write_db), durations(trace_db)
print(f"{'kernel':66s} {'launches/decode':>15s} {'us/decode':>10s} {'fetch GB (x2)':>14s} {'write GB':>9s} {'TB/s':>6s}")
tot_b = tot_us = 0.0
for k in sorted(d, key=lambda k: -d[k][0]):
    us = d[k][0] / ndec
    fb = f.get(k, [0, 0])[0] * 2 * 1024 / ndec
    wb = w.get(k, [0, 0])[0] * 1024 / ndec
    tot_b += fb + wb
    tot_us += us
    if us < 20:
        continue
    print(f"{k:66s} {d[k][1] / ndec:15.1f} {us:10.1f} {fb / 1e9:14.3f} {wb / 1e9:9.3f} {(fb + wb) / us / 1e6:6.2f}")
print(f"total: {tot_us / 1e3:.2f} ms of kernels per decode, {tot_b / 1e9:.2f} GB of HBM traffic -> {tot_b / tot_us / 1e6:.2f} TB/s average")
