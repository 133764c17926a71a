#!/bin/bash
# debugging aid (round 6, call 24): ptts_lm.hip compiled with one non-FULL strip-kernel family on the preloaded entry point, linked with the product's other objects
set -e
cd "$(dirname "$0")/.."
TORCH_LIB=$(python -c 'import torch, os; print(os.path.join(os.path.dirname(torch.__file__), "lib"))')
C=parler_tts_amd/csrc
build() {  # name, expression
  d=tools/variants/$1; mkdir -p $d
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-kernarg-preload-count=14 "-DPTTS_STRIP_PRELOAD(PRO,EPI,FULL)=$2" -c $C/ptts_lm.hip -o $d/ptts_lm.o
  g++ -shared -o $d/libptts_hip_$1.so $d/ptts_lm.o $C/ptts_lm_w8.o $C/ptts_dac.o $C/ptts_gemv_bf16.o $C/ptts_gemv_w8.o $C/ptts_gemv_f32.o $C/ptts_t5.o -L"$TORCH_LIB" -l:libamdhip64.so -Wl,-rpath,"$TORCH_LIB"
  echo built $d
}
build bis_plain "(FULL||PRO==0)" &
build bis_attn "(FULL||PRO==2)" &
build bis_lngelu "(FULL||(PRO==1&&EPI==1))" &
build bis_copy "(FULL||PRO==3)" &
build bis_lnstore "(FULL||(PRO==1&&EPI==0))" &
wait
