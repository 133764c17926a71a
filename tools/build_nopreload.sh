#!/bin/bash
# A/B build (round 5, kernarg preload): the decoder-LM translation units compiled WITHOUT -mllvm -amdgpu-kernarg-preload-count (the scalar arguments of
# the step's nodes then arrive by s_load like the by-value structs before) into tools/nopre/libptts_hip_nopre.so + tools/nopre/cabi_probe_nopre.
# The product library is not touched. Run on the GPU box: tools/nopre/cabi_probe_nopre lm 1
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/nopre
TORCH_LIB=$(python -c 'import torch, os; print(os.path.join(os.path.dirname(torch.__file__), "lib"))')
pids=()
for f in ptts_lm ptts_lm_w8 ptts_gemv_bf16 ptts_gemv_w8 ptts_gemv_f32; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -c parler_tts_amd/csrc/$f.hip -o tools/nopre/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
g++ -shared -o tools/nopre/libptts_hip_nopre.so tools/nopre/ptts_lm.o tools/nopre/ptts_lm_w8.o tools/nopre/ptts_gemv_bf16.o tools/nopre/ptts_gemv_w8.o \
    tools/nopre/ptts_gemv_f32.o -L"$TORCH_LIB" -l:libamdhip64.so -Wl,-rpath,"$TORCH_LIB"
hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude tools/cabi_probe.hip -o tools/nopre/cabi_probe_nopre -Ltools/nopre -lptts_hip_nopre \
    -Wl,-rpath,'$ORIGIN' -Wl,--unresolved-symbols=ignore-all
echo "built tools/nopre/cabi_probe_nopre"
