#!/bin/bash
# Measurement build: the decoder-LM translation units compiled with -DPTTS_TIMING (s_memtime phase stamps in every node of the
# single-utterance step, ptts_gemv.h: GV_STAMP) into tools/stamps/libptts_hip_stamps.so + tools/stamps/cabi_probe_stamps. The product library is
# NOT touched (no stamp code is compiled into it). Run on the GPU box: tools/stamps/cabi_probe_stamps lm 1
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/stamps
TORCH_LIB=$(python -c 'import torch, os; print(os.path.join(os.path.dirname(torch.__file__), "lib"))')
pids=()
for f in ptts_lm ptts_lm_w8 ptts_gemv_bf16 ptts_gemv_w8 ptts_gemv_f32; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DPTTS_TIMING -mllvm -amdgpu-kernarg-preload-count=14 -c parler_tts_amd/csrc/$f.hip -o tools/stamps/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
g++ -shared -o tools/stamps/libptts_hip_stamps.so tools/stamps/ptts_lm.o tools/stamps/ptts_lm_w8.o tools/stamps/ptts_gemv_bf16.o tools/stamps/ptts_gemv_w8.o \
    tools/stamps/ptts_gemv_f32.o -L"$TORCH_LIB" -l:libamdhip64.so -Wl,-rpath,"$TORCH_LIB"
hipcc --offload-arch=gfx950 -O2 -std=c++17 -DPTTS_STAMPS -Iinclude tools/cabi_probe.hip -o tools/stamps/cabi_probe_stamps -Ltools/stamps -lptts_hip_stamps \
    -Wl,-rpath,'$ORIGIN' -Wl,--unresolved-symbols=ignore-all
echo "built tools/stamps/cabi_probe_stamps"
