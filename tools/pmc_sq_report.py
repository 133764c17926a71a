"""Per-kernel wave-state breakdown from ONE rocprofv3 pass with SQ counters (8 SQ slots per pass on gfx950):
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
              --kernel-trace -d DIR -o p -- tools/cabi_probe dac 32 reps=2
    python tools/pmc_sq_report.py DIR/.../p_results.db
WAIT_ANY (parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stall: MFMA RAW / pipe busy) + ACTIVE_INST_ANY ~ WAVE_CYCLES (disjoint,
/opt/skills/guides/MI355X_MICROARCH.md section rocprofv3 PMC slots). Kernels are keyed by name + grid (anonymous-namespace names are stripped by the tool)."""
import re, sqlite3, sys
from collections import defaultdict

cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)").fetchall()]
name_col = next((c for c in ("counter_name", "name", "counter") if c in cols), None)
if name_col is None:
    print("counters_collection columns:", cols)
    sys.exit(1)
grid = "grid_size_x" if "grid_size_x" in cols else ("grid_x" if "grid_x" in cols else None)
lds = "lds_block_size" if "lds_block_size" in cols else None
sel = f"kernel_name, {name_col}, value" + (f", {grid}" if grid else ", 0") + (f", {lds}" if lds else ", 0")
acc, n = defaultdict(lambda: defaultdict(float)), defaultdict(int)
for k, c, v, g, l in cur.execute(f"select {sel} from counters_collection").fetchall():
    key = (re.sub(r"void |\(.*", "", k.replace("(anonymous namespace)::", ""))[:44], g, l)
    acc[key][c] += v
    if c == "SQ_WAVE_CYCLES":
        n[key] += 1
names = sorted({c for a in acc.values() for c in a})
print("counters:", " ".join(names))
print(f"{'kernel':44s} {'grid':>10s} {'lds':>7s} {'n':>4s} {'wave_cyc':>10s} {'parked':>7s} {'issue-stall':>11s} {'active':>7s} {'lds-stall':>9s} {'lds-active':>10s} {'bank-conf/lds-cyc':>17s}")
for key in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0))[: int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    a = acc[key]
    w = a.get("SQ_WAVE_CYCLES", 0.0)
    if w <= 0:
        continue
    pct = lambda c: f"{a.get(c, 0.0) / w * 100:6.1f}%"  # noqa: E731
    bc = a.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(a.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0)
    print(f"{key[0]:44s} {key[1]:10d} {key[2]:7d} {n[key]:4d} {w:10.3e} {pct('SQ_WAIT_ANY'):>7s} {pct('SQ_WAIT_INST_ANY'):>11s} {pct('SQ_ACTIVE_INST_ANY'):>7s} "
          f"{pct('SQ_WAIT_INST_LDS'):>9s} {pct('SQ_ACTIVE_INST_LDS'):>10s} {bc * 100:16.1f}%")
