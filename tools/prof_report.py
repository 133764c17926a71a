"""Summarise a rocprofv3 --kernel-trace sqlite DB: per-kernel table + one decode-step timeline."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select name, grid_x, grid_y, grid_z, workgroup_x, count(*), avg(duration), min(duration), max(duration), sum(duration), max(vgpr_count), max(lds_size) from kernels group by name, grid_x, grid_y, grid_z order by sum(duration) desc").fetchall()
tot = sum(r[9] for r in rows)
print(f"{'kernel':58s} {'grid':>16s} {'wg':>5s} {'calls':>7s} {'avg_us':>8s} {'min_us':>8s} {'max_us':>8s} {'share':>6s} {'vgpr':>5s} {'lds':>7s}")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    nm = re.sub(r'void |\(.*', '', r[0])[:58]
    print(f"{nm:58s} {str((r[1],r[2],r[3])):>16s} {r[4]:5d} {r[5]:7d} {r[6]/1e3:8.2f} {r[7]/1e3:8.2f} {r[8]/1e3:8.2f} {r[9]/tot*100:5.1f}% {r[10]:5d} {r[11]:7d}")
k = cur.execute("select name, start, end from kernels order by start").fetchall()
last = k[-196:]
span = (last[-1][2] - last[0][1]) / 1e3
busy = sum(x[2] - x[1] for x in last) / 1e3
print(f"last 196 dispatches (~1 decode step): span {span:.1f} us, sum of kernel durations {busy:.1f} us")
if len(sys.argv) > 3:  # decode steps in the trace: per-step sum over the kernels that run in every step (calls >= steps)
    steps = int(sys.argv[3])
    per = sum(r[9] for r in rows if r[5] >= steps) / steps / 1e3
    print(f"decode-step kernels (calls >= {steps}): sum of durations per step = {per:.1f} us (profiled: each dispatch carries ~0.9 us of tracing overhead)")
