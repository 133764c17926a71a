"""CPU: libptts_hip.so loads, exports every symbol include/ptts.h declares, and argument validation maps
error classes onto the reference's Python exceptions — no compute calls (no GPU here)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def _lib():
    from parler_tts_amd import _native as N

    if not os.path.exists(N.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    return N, N.load_library()


def test_exports_every_declared_symbol():
    N, lib = _lib()
    hdr = open(os.path.join(ROOT, "include", "ptts.h")).read()
    declared = set(re.findall(r"\b(ptts_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"ptts_engine", "ptts_dac", "ptts_t5"}
    assert declared == set(N.SYMBOLS), declared ^ set(N.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ptts_abi_version() == N.ABI_VERSION == 8


def test_invalid_config_is_value_error_not_crash():
    N, lib = _lib()
    cfg = N.PttsConfig(1000, 2, 16, 256, 9, 1088, 256, 0, 10000.0, 1024, 1024, 1025, N.PTTS_BF16, 1, 64, 16, 8, 0, 0, 0)  # 1000 % 16 != 0
    h = C.c_void_p()
    rc = lib.ptts_engine_create(C.byref(cfg), C.byref(h))
    assert rc == N.PTTS_E_INVALID
    with pytest.raises(ValueError, match="hidden_size"):
        N.check(rc, "ptts_engine_create")
    cfg = N.PttsConfig(1024, 2, 8, 256, 9, 1088, 256, 0, 10000.0, 1024, 1024, 1025, N.PTTS_BF16, 1, 64, 16, 8, 0, 0, 0)  # head_dim 128
    rc = lib.ptts_engine_create(C.byref(cfg), C.byref(h))
    with pytest.raises(NotImplementedError, match="head_dim"):
        N.check(rc)
    assert lib.ptts_engine_create(None, C.byref(h)) == N.PTTS_E_INVALID


def test_missing_library_fails_loudly(tmp_path):
    from parler_tts_amd import _native as N

    with pytest.raises(N.NativeLibraryError, match="no CPU fallback"):
        N.load_library(str(tmp_path / "nope.so"))


def test_engines_refuse_to_run_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from parler_tts_amd import _native as N
    from parler_tts_amd.engine import DacEngine, DecoderEngine

    with pytest.raises(N.NativeLibraryError, match="no CPU fallback"):
        DecoderEngine(hidden_size=128, num_layers=1, num_heads=2, ffn_dim=256, num_codebooks=9, vocab_size=1088, max_positions=64)
    with pytest.raises(N.NativeLibraryError, match="no CPU fallback"):
        DacEngine()
    from parler_tts_amd.engine import T5Engine

    with pytest.raises(N.NativeLibraryError, match="no CPU fallback"):
        T5Engine(vocab_size=100, d_model=128, d_kv=64, d_ff=256, num_layers=1, num_heads=2)


def test_header_is_plain_c_and_struct_layouts_match_the_ctypes_binding(tmp_path):
    """include/ptts.h is the drop-in boundary: it must compile as C99 (no torch / C++ types in the signatures) and as C++, and the
    struct layouts a C caller sees must be the ones parler_tts_amd/_native.py declares to ctypes (a silent mismatch would corrupt
    every configuration field)."""
    import shutil
    import subprocess

    N, lib = _lib()
    inc = os.path.join(ROOT, "include")
    gcc, gxx = shutil.which("gcc"), shutil.which("g++")
    if gcc is None or gxx is None:
        pytest.skip("no host compiler")
    subprocess.check_call([gcc, "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(inc, "ptts.h")])
    subprocess.check_call([gxx, "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", os.path.join(inc, "ptts.h")])
    src = tmp_path / "abi.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "ptts.h"\n'
                   'int main(void) {\n'
                   '  printf("%d %zu %zu %zu %zu %zu %zu %zu %zu\\n", PTTS_ABI_VERSION, sizeof(ptts_config), sizeof(ptts_gen_params), sizeof(ptts_dac_config),\n'
                   '         offsetof(ptts_config, rope_theta), offsetof(ptts_gen_params, seed), offsetof(ptts_dac_config, compute_dtype),\n'
                   '         sizeof(ptts_t5_config), offsetof(ptts_t5_config, layer_norm_eps));\n'
                   '  return 0;\n}\n')
    exe = tmp_path / "abi"
    subprocess.check_call([gcc, "-std=c99", "-I", inc, str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    assert [int(x) for x in out] == [N.ABI_VERSION, C.sizeof(N.PttsConfig), C.sizeof(N.PttsGenParams), C.sizeof(N.PttsDacConfig),
                                     N.PttsConfig.rope_theta.offset, N.PttsGenParams.seed.offset, N.PttsDacConfig.compute_dtype.offset,
                                     C.sizeof(N.PttsT5Config), N.PttsT5Config.layer_norm_eps.offset]


def test_torch_free_cxx_client_of_the_header_builds_and_links(tmp_path):
    """tools/cabi_probe.hip is a plain C++ client of include/ptts.h (engine create, weight load by the reference's tensor names, prefill,
    graph-replayed decode steps, DAC decode - device pointers and sizes only, no torch): it must compile against the header and link
    against libptts_hip.so as the header's consumers would. Without arguments it prints its usage before touching the GPU."""
    import shutil
    import subprocess

    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    N, _ = _lib()
    exe = str(tmp_path / "cabi_probe")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "cabi_probe.hip"),
           "-o", exe, "-L" + os.path.dirname(N.LIB_PATH), "-lptts_hip", "-Wl,-rpath," + os.path.dirname(N.LIB_PATH)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "usage:" in r.stderr, (r.returncode, r.stderr[-500:])
    # its `cmp` mode (A/B of two dumps: logits blocks + free-running ids) runs on the host
    import numpy as np

    def dump(path, logits, ids):
        with open(path, "wb") as f:
            np.array([0x70747473, logits.shape[1], logits.shape[2], logits.shape[0]], dtype=np.int64).tofile(f)
            logits.astype(np.float32).tofile(f)
            np.array(ids.shape, dtype=np.int64).tofile(f)
            ids.astype(np.int64).tofile(f)

    rng = np.random.default_rng(0)
    lg, ids = rng.standard_normal((3, 18, 40)), rng.integers(0, 1024, (18, 12))
    lg2, ids2 = lg.copy(), ids.copy()
    lg2[1, 4, :] = -lg2[1, 4, :]  # one row's arg-max moves
    ids2[3, 7] += 1
    a, b, c = (str(tmp_path / n) for n in ("a.bin", "b.bin", "c.bin"))
    dump(a, lg, ids), dump(b, lg, ids), dump(c, lg2, ids2)
    r = subprocess.run([exe, "cmp", a, b], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "max |a - b| = 0 " in r.stdout and "identical" in r.stdout, (r.returncode, r.stdout, r.stderr)
    r = subprocess.run([exe, "cmp", a, c], capture_output=True, text=True, timeout=60)
    assert r.returncode == 4 and "1 arg-max flips of 54 rows" in r.stdout and "differ from column 7" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_step_nodes_are_built_for_kernarg_preload(tmp_path):
    """The nodes of the single-utterance decode step (gemv_kernel / qkv_attn_kernel / xfold_attn_kernel) and, since round 6, the weight nodes of the batch > 8
    step and of a short prompt's prefill (gemm_strip_kernel's FULL instances, lnproj_fused_kernel, xattn_fused_kernel) take the first 56 bytes of their argument
    struct as scalars and the library is built with -mllvm -amdgpu-kernarg-preload-count=14 (-3.3 % per step). What that looks like
    in the code object: a kernel with preloaded arguments starts with the 256-byte compatibility prologue for firmware without the feature -
    s_load of exactly those arguments, s_waitcnt, s_branch to the real entry - which a kernel without preload never has. Checked on the embedded
    gfx950 code objects (no GPU needed); also: no pointer lost its address space on the way through the scalar parameters (no flat_load)."""
    import shutil
    import subprocess

    N, _ = _lib()
    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(llvm, "llvm-objdump")):
        pytest.skip("no llvm-objdump in this image")
    lib = str(tmp_path / "lib.so")
    shutil.copy(N.LIB_PATH, lib)
    subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", lib], capture_output=True, cwd=str(tmp_path), check=True)
    objs = [str(tmp_path / f) for f in os.listdir(tmp_path) if "amdgcn" in f]
    assert objs, "no embedded gfx950 code objects found"
    seen, bad, strips, glds, ttft = 0, [], {}, 0, 0
    for obj in objs:
        syms = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--symbols", obj], capture_output=True, text=True).stdout
        if "gemv_kernel" not in syms and "qkv_attn_kernel" not in syms and "gemm_strip_kernel" not in syms:
            continue  # translation units without the step's nodes
        dis = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", "--no-show-raw-insn", obj], capture_output=True, text=True).stdout
        cur, head, flat = None, {}, {}
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                cur = m.group(1)
                head[cur], flat[cur] = [], 0
                continue
            t = line.split()
            if cur is None or not t:
                continue
            if len(head[cur]) < 8:
                head[cur].append(t[0])
            if t[0].startswith("flat_load") or t[0].startswith("flat_store"):
                flat[cur] += 1
        for sym, ops in head.items():
            # round 6: the strip GEMM of the batch > 8 step has two entry points - the FULL instances (released checkpoint widths) on the preloaded
            # one (+ the non-FULL instances on prepared rows: T5's wo projection), everything else by value (ptts_lm_kernels.h: PTTS_STRIP_PRELOAD)
            ms = re.match(r"_Z(17gemm_strip_kernel|20gemm_strip_kernel_bv)I[tf]Li(\d+)ELi\d+ELi\d+ELb([01])E", sym)
            if ms:
                strips[ms.group(1)] = strips.get(ms.group(1), 0) + 1
                preload_expected = ms.group(3) == "1" or ms.group(2) == "3"  # FULL, or prepared rows (PRO_COPY = 3)
                if (ms.group(1) == "17gemm_strip_kernel") != preload_expected:
                    bad.append((sym[:70], "strip instance on the wrong entry point", ms.groups()))
                if ms.group(1) == "20gemm_strip_kernel_bv" and "s_branch" in ops:
                    bad.append((sym[:70], "by-value entry point with a preload prologue", ops))
            if not re.search(r"11gemv_kernel|15qkv_attn_kernel|17xfold_attn_kernel|_Z17gemm_strip_kernelI|_Z19lnproj_fused_kernelI|_Z18xattn_fused_kernelI|16gemm_glds_kernelI|_Z16rows_prep_kernelI|_Z19prefill_attn_kernelI|_Z24prefill_attn_mfma_kernelI|14t5_attn_kernelI|19t5_attn_mfma_kernelI", sym):
                continue
            seen += 1
            ttft += bool(re.search(r"rows_prep_kernelI|prefill_attn_kernelI|prefill_attn_mfma_kernelI|t5_attn_kernelI|t5_attn_mfma_kernelI", sym))  # call 54: the rest of the TTFT path
            glds += "16gemm_glds_kernelI" in sym  # round 6, call 52: the > 256-row LDS-DMA GEMM (time-to-first-token path) takes its first 56 bytes preloaded too
            if "s_branch" not in ops or not ops[0].startswith("s_load"):
                bad.append((sym[:60], "no preload prologue", ops))
            # (the batched cross K/V instances, PRO_COPY + EPI_KV, read their W / cache pointers from a device table: generic pointers by construction)
            if flat[sym] and not re.match(r"_Z17gemm_strip_kernelI[tf]Li3ELi3E", sym) and "16gemm_glds_kernelILi3E" not in sym:
                bad.append((sym[:60], "flat memory instructions", flat[sym]))
    assert seen >= 20, seen
    assert glds >= 10, glds
    assert ttft >= 20, ttft
    assert strips.get("17gemm_strip_kernel", 0) >= 20 and strips.get("20gemm_strip_kernel_bv", 0) >= 10, strips
    assert not bad, bad[:5]


def test_mfma_attention_kernels_request_their_lds_fragments_ahead_of_the_mfmas(tmp_path):
    """prefill_attn_mfma_kernel / t5_attn_mfma_kernel (the attention nodes of the time-to-first-token path above 128 (utterance, head) pairs): every K / V
    fragment of a key block is requested from LDS before the MFMAs that consume it (ptts_common.h: attn_block_scores / _v_request / _pv; DESIGN.md §4.4 (h):
    the compiler had scheduled ds_read -> s_waitcnt lgkmcnt(0) -> MFMA 16 + 64 times in a row, 14.3 -> 11.7 us per launch once removed). What that looks like
    in the shipped gfx950 code objects: at most a handful of `s_waitcnt lgkmcnt(0)` that follow one or two ds_reads ("short groups": one dependent LDS round trip
    each; tools/isa_lds_chains.py counts the same thing on the assembly), V read with b128 (no ds_read_b32 feeding the P V MFMAs), 128 MFMAs per key block."""
    import shutil
    import subprocess

    N, _ = _lib()
    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(llvm, "llvm-objdump")):
        pytest.skip("no llvm-objdump in this image")
    lib = str(tmp_path / "lib.so")
    shutil.copy(N.LIB_PATH, lib)
    subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", lib], capture_output=True, cwd=str(tmp_path), check=True)
    objs = [str(tmp_path / f) for f in os.listdir(tmp_path) if "amdgcn" in f]
    assert objs, "no embedded gfx950 code objects found"
    seen, bad = 0, []
    for obj in objs:
        syms = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--symbols", obj], capture_output=True, text=True).stdout
        if "attn_mfma_kernel" not in syms:
            continue
        dis = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", "--no-show-raw-insn", obj], capture_output=True, text=True).stdout
        cur, stat = None, {}
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                cur = m.group(1) if "attn_mfma_kernel" in m.group(1) else None
                if cur:
                    stat[cur] = dict(reads=0, b32=0, short=0, since=0, mfma=0)
                continue
            t = line.split()
            if cur is None or not t:
                continue
            s = stat[cur]
            if t[0].startswith("ds_read"):
                s["reads"] += 1
                s["since"] += 1
                s["b32"] += t[0] in ("ds_read_b32", "ds_read2_b32", "ds_read2st64_b32")
            elif t[0].startswith("v_mfma"):
                s["mfma"] += 1
            elif t[0] == "s_waitcnt" and "lgkmcnt(" in line:
                if 0 < s["since"] <= 2 and "lgkmcnt(0)" in line:
                    s["short"] += 1
                s["since"] = 0
        for sym, s in stat.items():
            seen += 1
            if s["mfma"] < 128 or s["mfma"] % 128:
                bad.append((sym[:60], "MFMAs per key block", s))
            if s["short"] > 6:
                bad.append((sym[:60], "LDS reads waited for one at a time", s))
            if s["b32"]:
                bad.append((sym[:60], "V read element by element", s))
    assert seen >= 10, seen  # prefill: 1..4 waves x {bf16, fp32}; T5: {bf16, fp32}
    assert not bad, bad[:5]
