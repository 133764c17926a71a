"""Which kernels wait for their LDS reads one at a time? (round 6, call 46: the f32-MFMA attention kernels had every K / V fragment read compiled as
ds_read -> s_waitcnt lgkmcnt(0) -> MFMA, 16 + 64 LDS round trips in a row per key block.) Reads the device assembly of a translation unit
(hipcc -S --cuda-device-only) and prints per kernel: ds_read instructions, s_waitcnt lgkmcnt(0) instructions that follow at most two ds_reads issued
since the previous lgkmcnt wait ("short groups": each is one dependent LDS round trip, ~64-128 cycles), MFMA count.
   python tools/isa_lds_chains.py file.s [min_short_groups]"""
import re
import subprocess
import sys

text = open(sys.argv[1]).read()
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = []
for m in re.finditer(r"^(_Z\w+):[^\n]*\n", text, re.M):
    name, i = m.group(1), m.end()
    j = text.find("s_endpgm", i)
    if j < 0:
        continue
    reads = short = since = mfma = 0
    for line in text[i:j].split("\n"):
        t = line.strip()
        if t.startswith("ds_read") or t.startswith("ds_bpermute") or t.startswith("ds_swizzle"):
            reads += 1
            since += 1
        elif t.startswith("v_mfma"):
            mfma += 1
        elif t.startswith("s_waitcnt") and "lgkmcnt(" in t:
            if 0 < since <= 2 and "lgkmcnt(0)" in t:
                short += 1
            since = 0
    if short >= thr:
        rows.append((short, reads, mfma, name))
names = subprocess.run(["c++filt"] + [r[3] for r in rows], capture_output=True, text=True).stdout.split("\n") if rows else []
for (short, reads, mfma, _), dn in sorted(zip(rows, names), reverse=True):
    print(f"{short:4d} short LDS groups  {reads:4d} ds reads  {mfma:4d} mfma   {dn[:160]}")
