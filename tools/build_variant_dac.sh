#!/bin/bash
# A/B build of the codec: ptts_dac.hip compiled with extra -D flags (e.g. -DPTTS_DEV_KNOBS) and linked with the product's other objects into
# tools/variants/<name>/libptts_hip_<name>.so + cabi_probe_<name>. The product library is not touched (run __graft_entry__.build() first: its objects are used).
set -e
name=$1; shift
cd "$(dirname "$0")/.."
d=tools/variants/$name
mkdir -p $d
TORCH_LIB=$(python -c 'import torch, os; print(os.path.join(os.path.dirname(torch.__file__), "lib"))')
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-kernarg-preload-count=14 "$@" -c parler_tts_amd/csrc/ptts_dac.hip -o $d/ptts_dac.o
others=$(ls parler_tts_amd/csrc/*.o | grep -v '/ptts_dac.o$')
g++ -shared -o $d/libptts_hip_$name.so $d/ptts_dac.o $others -L"$TORCH_LIB" -l:libamdhip64.so -Wl,-rpath,"$TORCH_LIB"
hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude tools/cabi_probe.hip -o $d/cabi_probe_$name -L$d -lptts_hip_$name -Wl,-rpath,'$ORIGIN'
echo "built $d/cabi_probe_$name"
