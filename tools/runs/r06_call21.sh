#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
echo "---- product (preloaded scalars)"; python tools/small_spec_probe.py 2>&1 | grep -v amdgpu.ids | grep bf16
echo "---- by-value variant"; PTTS_LIB=$PWD/tools/variants/byval/libptts_hip_byval.so python tools/small_spec_probe.py 2>&1 | grep -v amdgpu.ids | grep bf16
