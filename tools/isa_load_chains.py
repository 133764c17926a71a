"""Which kernels wait for their global loads one at a time? (round 6, call 38: prefill_attn_mfma_kernel's per-lane conditional loads had been compiled into
26 branch + load + s_waitcnt vmcnt(0) groups in a row.) Reads the device assembly of a translation unit (hipcc -S --cuda-device-only) and prints per
kernel: global / buffer loads, s_waitcnt vmcnt(...) instructions, and how many of the waits follow at most two loads issued since the previous wait
("short groups": each is one dependent memory round trip).   python tools/isa_load_chains.py file.s [min_short_groups]"""
import re
import subprocess
import sys

text = open(sys.argv[1]).read()
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rows = []
for m in re.finditer(r"^(_Z\w+):[^\n]*\n", text, re.M):
    name, i = m.group(1), m.end()
    j = text.find("s_endpgm", i)
    if j < 0:
        continue
    loads = waits = short = since = 0
    for line in text[i:j].split("\n"):
        t = line.strip()
        if re.match(r"(global_load_|buffer_load_|flat_load_)", t) and " lds" not in t:
            loads += 1
            since += 1
        elif t.startswith("s_waitcnt") and "vmcnt(" in t:
            waits += 1
            if 0 < since <= 2 and "vmcnt(0)" in t:  # a counted wait (vmcnt(N > 0)) leaves loads in flight: software pipelining, not a round trip
                short += 1
            since = 0
    if short >= thr:
        rows.append((short, loads, waits, name))
names = subprocess.run(["c++filt"] + [r[3] for r in rows], capture_output=True, text=True).stdout.split("\n") if rows else []
for (short, loads, waits, _), dn in sorted(zip(rows, names), reverse=True):
    print(f"{short:4d} short groups  {loads:4d} loads  {waits:4d} vmcnt waits   {dn[:150]}")
