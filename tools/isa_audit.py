#!/usr/bin/env python
"""Static resource audit of every gfx950 kernel in libptts_hip.so (no GPU needed): registers, LDS, scratch (spills),
occupancy bound, code size, and a few instruction counts from the disassembly (MFMA, LDS reads, global loads, waits).

  python tools/isa_audit.py [--lib parler_tts_amd/libptts_hip.so] [--isa] > profiles/r02_isa_resources.txt

What it is for: a kernel that spills (scratch > 0), falls off its intended occupancy, or carries an s_waitcnt vmcnt(0)
inside what should be a pipelined loop shows up here before any GPU time is spent. The numbers are the compiler's
(NT_AMDGPU_METADATA of the embedded code objects); the occupancy bound is the CDNA3/4 rule for wave64: 512 VGPRs per
SIMD lane (unified arch + acc file), allocation granule 8, at most 8 waves per SIMD, and the 160 KB LDS per CU.
"""
from __future__ import annotations

import argparse
import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    p = subprocess.run([shutil.which("c++filt") or os.path.join(LLVM, "llvm-cxxfilt")], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.strip().split("\n") if p.returncode == 0 else list(names)


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("unsigned short", "bf16").replace("unsigned char", "u8")
    name = re.sub(r"\((anonymous namespace::)?\w*Args\)$", "", name)
    name = re.sub(r"^void ", "", name)
    return name


def parse_metadata(text: str):
    """Very small YAML subset: the '- .key: value' records under amdhsa.kernels."""
    kernels, cur, in_args = [], None, False
    for line in text.splitlines():
        m = re.match(r"^  - \.(\w+):\s*(.*)$", line)
        if m:  # first key of a new kernel record
            cur = {m.group(1): m.group(2).strip()}
            kernels.append(cur)
            in_args = m.group(1) == "args"
            continue
        m = re.match(r"^    \.(\w+):\s*(.*)$", line)
        if m and cur is not None:
            cur[m.group(1)] = m.group(2).strip().strip("'")
            in_args = m.group(1) == "args"
    return [k for k in kernels if "name" in k]


def occupancy(vgpr: int, agpr: int, lds: int, wg_threads: int) -> int:
    """waves per SIMD the register file / LDS allow (the launch may ask for fewer)."""
    regs = max(1, -(-(vgpr + agpr) // 8) * 8)
    by_reg = min(8, 512 // regs)
    if lds <= 0 or wg_threads <= 0:
        return by_reg
    waves_per_wg = -(-wg_threads // 64)
    wgs_per_cu = max(1, (160 * 1024) // max(lds, 1))
    by_lds = max(1, (wgs_per_cu * waves_per_wg) // 4)
    return max(1, min(by_reg, by_lds))


def isa_counts(obj: str):
    """per-kernel instruction histogram from llvm-objdump -d"""
    p = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", obj], capture_output=True, text=True)
    out, cur = {}, None
    for line in p.stdout.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = collections.Counter()
            out[m.group(1)] = cur
            continue
        if cur is None:
            continue
        t = line.split()
        if not t:
            continue
        op = t[0]
        cur["insts"] += 1
        if op.startswith("v_mfma") or op.startswith("v_smfmac"):
            cur["mfma"] += 1
        elif op.startswith("ds_read") or op.startswith("ds_load"):
            cur["ds_read"] += 1
        elif op.startswith("ds_write") or op.startswith("ds_store"):
            cur["ds_write"] += 1
        elif op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load"):
            cur["vmem_ld"] += 1
        elif op.startswith("global_store") or op.startswith("buffer_store") or op.startswith("flat_store"):
            cur["vmem_st"] += 1
        elif op.startswith("global_atomic") or op.startswith("buffer_atomic") or op.startswith("flat_atomic"):
            cur["atomic"] += 1
        elif op.startswith("scratch_"):
            cur["scratch"] += 1
        elif op == "s_waitcnt":
            cur["waitcnt"] += 1
            if "vmcnt(0)" in line:
                cur["vmcnt0"] += 1
        elif op == "s_barrier":
            cur["barrier"] += 1
        elif op.startswith("v_dot2"):
            cur["dot2"] += 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "parler_tts_amd", "libptts_hip.so"))
    ap.add_argument("--isa", action="store_true", help="also disassemble and count instructions per kernel")
    ap.add_argument("--filter", default="", help="only kernels whose shortened demangled name matches this regular expression")
    args = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="isa_audit_")
    try:
        lib = os.path.join(tmp, "lib.so")
        shutil.copy(args.lib, lib)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", lib], capture_output=True, cwd=tmp)
        objs = sorted(f for f in os.listdir(tmp) if "amdgcn" in f)
        rows = []
        for f in objs:
            path = os.path.join(tmp, f)
            meta = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", path], capture_output=True, text=True).stdout
            ks = parse_metadata(meta)
            counts = isa_counts(path) if args.isa else {}
            for k in ks:
                sym = k["name"]
                rows.append((sym, k, counts.get(sym, {})))
        names = demangle([r[0] for r in rows])
        table = []
        for (sym, k, cnt), dn in zip(rows, names):
            dn = short(dn)
            if args.filter and not re.search(args.filter, dn):
                continue
            vg, ag, sg = int(k.get("vgpr_count", 0)), int(k.get("agpr_count", 0)), int(k.get("sgpr_count", 0))
            lds, scr = int(k.get("group_segment_fixed_size", 0)), int(k.get("private_segment_fixed_size", 0))
            wg = int(k.get("max_flat_workgroup_size", 0))
            table.append((dn, vg, ag, sg, lds, scr, wg, occupancy(vg, ag, lds, wg), int(k.get("vgpr_spill_count", 0)), int(k.get("sgpr_spill_count", 0)), cnt))
        table.sort(key=lambda r: r[0])
        print(f"# tools/isa_audit.py{' --isa' if args.isa else ''} on {os.path.relpath(args.lib, ROOT)} ({len(table)} gfx950 kernels in {len(objs)} code objects)")
        print("# vgpr = arch VGPRs (+agpr accumulation file), lds = static LDS bytes (dynamic LDS is set at launch), scratch = private segment bytes")
        print("# per lane (0 = no spills), occ = waves / SIMD the register file (512 / granule 8, <= 8) and the static LDS allow, wg = max workgroup threads")
        hdr = f"{'kernel':<86} {'vgpr':>4} {'agpr':>4} {'sgpr':>4} {'lds':>6} {'scratch':>7} {'wg':>5} {'occ':>3} {'vspill':>6} {'sspill':>6}"
        if args.isa:
            hdr += f" {'insts':>6} {'mfma':>5} {'dot2':>5} {'vm_ld':>5} {'vm_st':>5} {'ds_rd':>5} {'ds_wr':>5} {'atom':>4} {'wait':>4} {'vm0':>3} {'bar':>3}"
        print(hdr)
        spills = []
        for dn, vg, ag, sg, lds, scr, wg, occ, vs, ss, cnt in table:
            line = f"{dn[:86]:<86} {vg:>4} {ag:>4} {sg:>4} {lds:>6} {scr:>7} {wg:>5} {occ:>3} {vs:>6} {ss:>6}"
            if args.isa:
                line += (f" {cnt.get('insts', 0):>6} {cnt.get('mfma', 0):>5} {cnt.get('dot2', 0):>5} {cnt.get('vmem_ld', 0):>5} {cnt.get('vmem_st', 0):>5}"
                         f" {cnt.get('ds_read', 0):>5} {cnt.get('ds_write', 0):>5} {cnt.get('atomic', 0):>4} {cnt.get('waitcnt', 0):>4} {cnt.get('vmcnt0', 0):>3} {cnt.get('barrier', 0):>3}")
            print(line)
            if scr or vs or ss:
                spills.append(dn)
        print(f"# kernels with scratch / spills: {len(spills)}" + ("" if not spills else ": " + "; ".join(s[:60] for s in spills)))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
