#!/bin/bash
# First GPU call of the next round: every parked experiment measured in ONE gpurun call (~5 GPU-minutes, no compilation on the box).
#   bash tools/experimental/build_variants.sh                                   # here, on the CPU: one library per patch under parler_tts_amd/exp/
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/experimental/run_all.sh'
# Each variant: guarded parity tests (tools/experimental/test_experimental_gpu.py) -> probe, selected with PTTS_LIB. A variant whose
# library is missing is skipped (build it first). Results in gpurun_out/r03a/summary.txt.
set -u
O=gpurun_out/r03a; mkdir -p $O
S=$O/summary.txt; : > $S
E=parler_tts_amd/exp
step32() { local tag=$1; shift; env "$@" timeout 120 python tools/step_probe2.py 32 "$tag" 2>&1 | grep step_probe2 >> $S; }
tests() { local tag=$1 k=$2; shift 2; env "$@" timeout 400 python -m pytest tools/experimental/test_experimental_gpu.py -q -x -k "$k" > $O/pytest_$tag.log 2>&1; echo "[$tag] $(tail -1 $O/pytest_$tag.log)" >> $S; }

echo "## baseline (HEAD library)" >> $S
step32 baseline PTTS_NOOP=1
timeout 200 python tools/dac_probe.py 2>&1 | grep "B=1 T=860\|B=32" >> $S

echo "## batch-32 GEMM nodes in a cold dependent chain, phase stamps (tools/chain_probe32: build it on the CPU first)" >> $S
[ -x tools/chain_probe32 ] && timeout 120 tools/chain_probe32 >> $S 2>&1 || echo "tools/chain_probe32 missing" >> $S

echo "## decode at batch 64 / 128 per GPU (default library): parity vs the oracle, then the step time" >> $S
tests big "above_32"
for B in 64 128; do timeout 200 python tools/step_probe2.py $B b$B 2>&1 | grep "step_probe2\|Error\|error" | tail -2 >> $S; done

echo "## PTTS_DECODE_STREAMS (in the tree): e2e generate() of 32 utterances, 860 frames, 1 / 2 / 4 sub-batches" >> $S
cat > $O/e2e32.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda", 0)
m = bench.build_model(0, 1, dev, torch.bfloat16)
d, p = bench.synthetic_batch(32, 0, dev)
kw = dict(input_ids=d, prompt_input_ids=p, do_sample=False, max_new_tokens=bench.NEW_TOKENS, min_new_tokens=bench.NEW_TOKENS)
for n in (1, 2, 4):
    m.decode_streams = n
    m.generate(**kw); torch.cuda.synchronize()
    t0 = time.perf_counter(); m.generate(**kw); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"[e2e bs=32] decode_streams={n}: {dt * 1e3:.1f} ms per generate() = {32 * bench.AUDIO_S / dt:.1f} audio-s/s", flush=True)
# batch 8 as two GEMV-step sub-batches of 4 (decode_streams_min_sub = 4) against the single MFMA-strip engine
d, p = bench.synthetic_batch(8, 0, dev)
kw = dict(input_ids=d, prompt_input_ids=p, do_sample=False, max_new_tokens=bench.NEW_TOKENS, min_new_tokens=bench.NEW_TOKENS)
m.decode_streams_min_sub = 4
m._engine = None  # fresh engines for this leg
for n in (1, 2):
    m.decode_streams = n
    m.generate(**kw); torch.cuda.synchronize()
    t0 = time.perf_counter(); m.generate(**kw); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"[e2e bs=8] decode_streams={n}: {dt * 1e3:.1f} ms per generate() = {8 * bench.AUDIO_S / dt:.1f} audio-s/s", flush=True)
PY
timeout 400 python $O/e2e32.py 2>&1 | grep "e2e bs=" >> $S

for V in fc2_last_arriver xattn_groups lm_batch32_both; do
  echo "## $V" >> $S
  L=$E/libptts_$V.so
  [ -f $L ] || { echo "[$V] $L missing: run tools/experimental/build_variants.sh first" >> $S; continue; }
  case $V in
    fc2_last_arriver) K="arriver"; F="PTTS_FC2_ARRIVE=1";;
    xattn_groups)     K="groups";  F="PTTS_XATTN_GROUPS=1";;
    *)                K="together or arriver or groups"; F="PTTS_FC2_ARRIVE=1 PTTS_XATTN_GROUPS=1";;
  esac
  tests $V "$K" PTTS_LIB=$PWD/$L
  step32 $V PTTS_LIB=$PWD/$L $F
done

echo "## dac_fused_resunit" >> $S
L=$E/libptts_dac_fused_resunit.so
if [ -f $L ]; then
  tests dacfuse "fused_residual" PTTS_LIB=$PWD/$L
  PTTS_LIB=$PWD/$L PTTS_DAC_FUSE_RES=1 timeout 200 python tools/dac_probe.py 2>&1 | grep "B=1 T=860\|B=32" >> $S
else
  echo "[dacfuse] $L missing" >> $S
fi

echo "## bench.py --live-pmc (roofline.traffic measured in the run)" >> $S
timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --live-pmc 2>/dev/null | python -c "
import json, sys
j = json.loads(sys.stdin.readline()); r = j['roofline']; print('traffic', r['traffic'], '|', r['traffic_note'][:80])" >> $S 2>&1
cat $S
