"""Per-kernel MFMA utilisation from ONE rocprofv3 pass with the counters SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE:
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d DIR -o p -- python tools/dac_bf16_probe.py
    python tools/pmc_mfma_report.py DIR/.../p_results.db
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs) (gfx94x derived-counter formula; ROCm 7.2 ships no
gfx950 section, /opt/skills/guides/MI355X_MICROARCH.md §rocprofv3 PMC slots; GRBM_GUI_ACTIVE is summed over the 8 XCDs)."""
import re, sqlite3, sys
from collections import defaultdict

cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)").fetchall()]
name_col = next((c for c in ("counter_name", "name", "counter") if c in cols), None)
if name_col is None:
    print("counters_collection columns:", cols)
    sys.exit(1)
rows = cur.execute(f"select kernel_name, {name_col}, value from counters_collection").fetchall()
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for k, c, v in rows:
    key = re.sub(r"void |\(.*", "", k.replace("(anonymous namespace)::", ""))[:64]
    acc[key][c] += v
    if c == "GRBM_GUI_ACTIVE":
        cnt[key] += 1
print(f"{'kernel':66s} {'dispatches':>10s} {'GRBM_GUI_ACTIVE':>16s} {'MFMA_BUSY_CYCLES':>17s} {'MfmaUtil':>9s}")
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
    g, m = acc[k].get("GRBM_GUI_ACTIVE", 0.0), acc[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if m <= 0:
        continue
    print(f"{k:66s} {cnt[k]:10d} {g:16.3e} {m:17.3e} {m / (g / 8 * 256 * 4) * 100:8.1f}%")
