#!/bin/bash
# First GPU call of the next round: every parked experiment measured in ONE gpurun call (~6-8 GPU-minutes).
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/experimental/run_all.sh'
# For each patch: apply -> build -> guarded parity tests -> probe -> revert. Results in gpurun_out/r03a/summary.txt.
# The snapshot on the GPU box has no .git: `git apply` works on a plain directory.
set -u
O=gpurun_out/r03a; mkdir -p $O
S=$O/summary.txt; : > $S
build() { python -c "import __graft_entry__ as g; g.build()" > $O/build_$1.log 2>&1 && echo "[$1] build ok" >> $S || { echo "[$1] BUILD FAILED" >> $S; tail -5 $O/build_$1.log >> $S; return 1; }; }
step32() { local tag=$1; shift; env "$@" timeout 120 python tools/step_probe2.py 32 "$tag" 2>&1 | grep step_probe2 >> $S; }

echo "## baseline (HEAD)" >> $S
step32 baseline PTTS_NOOP=1
timeout 200 python tools/dac_probe.py 2>&1 | grep "B=1 T=860\|B=32" >> $S

echo "## PTTS_DECODE_STREAMS=2 (already in the tree): e2e generate() of 32 utterances, 860 frames" >> $S
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
PY
timeout 300 python $O/e2e32.py 2>&1 | grep "e2e bs=32" >> $S

for P in fc2_last_arriver xattn_groups; do
  echo "## $P" >> $S
  git apply tools/experimental/$P.patch || { echo "[$P] patch does not apply" >> $S; continue; }
  if build $P; then
    PTTS_TEST_EXPERIMENTAL=1 timeout 400 python -m pytest tests/test_lm_gpu.py -q -x -k "experimental or producer_statistics or batch_sizes" > $O/pytest_$P.log 2>&1; tail -1 $O/pytest_$P.log >> $S
    if [ $P = fc2_last_arriver ]; then step32 $P PTTS_FC2_ARRIVE=1; else step32 $P PTTS_XATTN_GROUPS=1; fi
  fi
  git apply -R tools/experimental/$P.patch
done

echo "## both LM patches" >> $S
git apply tools/experimental/lm_batch32_both.patch && build both && step32 both PTTS_FC2_ARRIVE=1 PTTS_XATTN_GROUPS=1
git apply -R tools/experimental/lm_batch32_both.patch

echo "## dac_fused_resunit" >> $S
git apply tools/experimental/dac_fused_resunit.patch && build dacfuse && {
  PTTS_TEST_EXPERIMENTAL=1 timeout 300 python -m pytest tests/test_dac_gpu.py -q -x -k "fused_residual or full_size" > $O/pytest_dacfuse.log 2>&1; tail -1 $O/pytest_dacfuse.log >> $S
  PTTS_DAC_FUSE_RES=1 timeout 200 python tools/dac_probe.py 2>&1 | grep "B=1 T=860\|B=32" >> $S
}
git apply -R tools/experimental/dac_fused_resunit.patch
build head   # leave the tree and the .so as they were
cat $S
