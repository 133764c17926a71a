#!/bin/bash
# A/B build: the decoder-LM translation units compiled with extra -D flags into tools/variants/<name>/libptts_hip_<name>.so + cabi_probe_<name>.
# The product library is not touched.   tools/build_variant.sh ks2 -DPTTS_FC2_KSPLIT=2 ; on the GPU box: tools/variants/ks2/cabi_probe_ks2 lm 32
set -e
name=$1; shift
cd "$(dirname "$0")/.."
d=tools/variants/$name
mkdir -p $d
TORCH_LIB=$(python -c 'import torch, os; print(os.path.join(os.path.dirname(torch.__file__), "lib"))')
pids=()
for f in ptts_lm ptts_lm_w8 ptts_gemv_bf16 ptts_gemv_w8 ptts_gemv_f32; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-kernarg-preload-count=14 "$@" -c parler_tts_amd/csrc/$f.hip -o $d/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
g++ -shared -o $d/libptts_hip_$name.so $d/ptts_lm.o $d/ptts_lm_w8.o $d/ptts_gemv_bf16.o $d/ptts_gemv_w8.o $d/ptts_gemv_f32.o -L"$TORCH_LIB" -l:libamdhip64.so -Wl,-rpath,"$TORCH_LIB"
hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude tools/cabi_probe.hip -o $d/cabi_probe_$name -L$d -lptts_hip_$name -Wl,-rpath,'$ORIGIN' -Wl,--unresolved-symbols=ignore-all
echo "built $d/cabi_probe_$name"
