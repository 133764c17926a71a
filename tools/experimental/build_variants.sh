#!/bin/bash
# CPU-side (no GPU needed): build every parked patch into its OWN shared library under parler_tts_amd/exp/, so that the GPU call measures
# them by switching PTTS_LIB instead of re-compiling on the box (a ptts_lm.hip build is ~1 GPU-minute; four of them are 5 % of a round's budget).
#   bash tools/experimental/build_variants.sh          # from the repo root, after `python -c "import __graft_entry__ as g; g.build()"`
# The tree is not touched: each patch is applied to a scratch copy of the sources; only the translation unit it changes is recompiled
# and linked with the tree's other objects. *.so are git-ignored but travel with the gpurun snapshot.
set -eu
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd "$ROOT"
python -c "import __graft_entry__ as g; g.build()"   # the tree's own objects / library first
TORCH_LIB=$(python -c "import os, torch; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
mkdir -p parler_tts_amd/exp
for V in fc2_last_arriver xattn_groups lm_batch32_both dac_fused_resunit; do
  T=$(mktemp -d /tmp/ptts_variant_XXXX)
  mkdir -p "$T/parler_tts_amd" "$T/include"
  cp -r parler_tts_amd/csrc "$T/parler_tts_amd/"; cp include/ptts.h "$T/include/"
  (cd "$T" && git apply --include='parler_tts_amd/*' "$ROOT/tools/experimental/$V.patch")
  if [ "$V" = dac_fused_resunit ]; then TU=ptts_dac; else TU=ptts_lm; fi
  (cd "$T/parler_tts_amd/csrc" && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed -c $TU.hip -o $TU.o)
  OBJS=""
  for O in ptts_lm ptts_dac ptts_gemv_bf16 ptts_gemv_w8 ptts_gemv_f32; do
    if [ "$O" = "$TU" ]; then OBJS="$OBJS $T/parler_tts_amd/csrc/$O.o"; else OBJS="$OBJS $ROOT/parler_tts_amd/csrc/$O.o"; fi
  done
  g++ -shared -o "parler_tts_amd/exp/libptts_$V.so" $OBJS -L"$TORCH_LIB" -l:libamdhip64.so -Wl,-rpath,"$TORCH_LIB"
  rm -rf "$T"
  echo "built parler_tts_amd/exp/libptts_$V.so"
done
hipcc --offload-arch=gfx950 -O3 -w -std=c++17 -o tools/chain_probe32 tools/chain_probe32.hip && echo "built tools/chain_probe32"
python - <<'PY'
import ctypes, glob, os, torch  # noqa: F401  (torch first: one HIP runtime instance)
for p in sorted(glob.glob("parler_tts_amd/exp/libptts_*.so")):
    lib = ctypes.CDLL(os.path.abspath(p)); print(p, "loads, ABI", lib.ptts_abi_version())
PY
