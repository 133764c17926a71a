#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python tools/small_spec_probe.py 2>&1 | grep -v amdgpu.ids
echo ---- PTTS_NO_FO_PREFILL=1; PTTS_NO_FO_PREFILL=1 python tools/small_spec_probe.py 2>&1 | grep -v amdgpu.ids
echo ---- PTTS_NO_KV_BATCHED=1; PTTS_NO_KV_BATCHED=1 python tools/small_spec_probe.py 2>&1 | grep -v amdgpu.ids
