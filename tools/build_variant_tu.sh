#!/bin/bash
# A/B build of chosen translation units: tools/build_variant_tu.sh <name> <tu[,tu...]> [-D flags] compiles parler_tts_amd/csrc/<tu>.hip with the extra
# flags (e.g. -DPTTS_DEV_KNOBS) and links them with the product's other objects into tools/variants/<name>/libptts_hip_<name>.so + cabi_probe_<name>.
# The product library is not touched (run __graft_entry__.build() first: its objects are used).
#   tools/build_variant_tu.sh dacdev ptts_dac -DPTTS_DEV_KNOBS        tools/build_variant_tu.sh t5dev ptts_t5,ptts_lm -DPTTS_DEV_KNOBS
set -e
name=$1; tus=$2; shift 2
cd "$(dirname "$0")/.."
d=tools/variants/$name
mkdir -p $d
TORCH_LIB=$(python -c 'import torch, os; print(os.path.join(os.path.dirname(torch.__file__), "lib"))')
pids=(); mine=(); others=$(ls parler_tts_amd/csrc/*.o)
for tu in ${tus//,/ }; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-kernarg-preload-count=14 "$@" -c parler_tts_amd/csrc/$tu.hip -o $d/$tu.o &
  pids+=($!); mine+=($d/$tu.o); others=$(echo "$others" | grep -v "/$tu.o\$")
done
for p in "${pids[@]}"; do wait $p; done
g++ -shared -o $d/libptts_hip_$name.so "${mine[@]}" $others -L"$TORCH_LIB" -l:libamdhip64.so -Wl,-rpath,"$TORCH_LIB"
hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude tools/cabi_probe.hip -o $d/cabi_probe_$name -L$d -lptts_hip_$name -Wl,-rpath,'$ORIGIN'
echo "built $d/cabi_probe_$name"
