#!/bin/bash
# round 6, GPU call 24: which non-FULL strip family breaks on the preloaded entry point (small golden spec, bf16 decode)? + FULL instances of the
# same families at Mini-v1 widths (PTTS_NO_GEMV=1, 2 utterances) against the GEMV step
cd "$GRAFT_REPO_ROOT" || exit 1
{
echo "---- product (FULL preloaded, non-FULL by value)"; python tools/small_spec_probe.py 2>&1 | grep -v amdgpu.ids | grep bf16
for v in bis_plain bis_attn bis_lngelu bis_copy bis_lnstore; do
  echo "---- $v"; PTTS_LIB=$PWD/tools/variants/$v/libptts_hip_$v.so python tools/small_spec_probe.py 2>&1 | grep -v amdgpu.ids | grep "bf16\|Error" | tail -5
done
echo "---- FULL instances at 2 / 6 utterances: GEMV step vs strips (PTTS_NO_GEMV=1)"
for B in 2 6; do
  tools/cabi_probe lm $B dump=/tmp/a$B.bin steps=6 tag=gemv 2>&1 | grep -v amdgpu.ids | cut -c1-100
  PTTS_NO_GEMV=1 tools/cabi_probe lm $B dump=/tmp/b$B.bin steps=6 tag=strips 2>&1 | grep -v amdgpu.ids | cut -c1-100
  tools/cabi_probe cmp /tmp/a$B.bin /tmp/b$B.bin 2>&1 | tail -3
done
} | tee gpurun_out/r06_strip_preload_bisect.txt
