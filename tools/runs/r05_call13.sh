#!/bin/bash
# round 5, GPU call 13: tiled cross-attention fold kernels (bit-identity vs the row kernels + kernel times), DAC switches (k1 convs / last transposed conv on the
# LDS-tiled kernel), the final conv + tanh stage pin, folded-path LM tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
{
timeout 120 tools/cabi_probe lm 1 tag=fold_tiled dump=/tmp/ft.bin steps=4
PTTS_FOLD_TILED=0 timeout 120 tools/cabi_probe lm 1 tag=fold_rows dump=/tmp/fr.bin steps=4
tools/cabi_probe cmp /tmp/ft.bin /tmp/fr.bin | tail -2
timeout 120 tools/cabi_probe lm 1 fp32 tag=fold_tiled dump=/tmp/ft32.bin steps=4
PTTS_FOLD_TILED=0 timeout 120 tools/cabi_probe lm 1 fp32 tag=fold_rows dump=/tmp/fr32.bin steps=4
tools/cabi_probe cmp /tmp/ft32.bin /tmp/fr32.bin | tail -2
timeout 120 tools/cabi_probe lm 1 large fp8 tag=fold_tiled dump=/tmp/ft8.bin steps=4
PTTS_FOLD_TILED=0 timeout 120 tools/cabi_probe lm 1 large fp8 tag=fold_rows dump=/tmp/fr8.bin steps=4
tools/cabi_probe cmp /tmp/ft8.bin /tmp/fr8.bin | tail -2
for B in 32 1; do
  timeout 120 tools/cabi_probe dac $B tag=default
  PTTS_DAC_LDS_K1=1 timeout 120 tools/cabi_probe dac $B tag=k1_and_last_up_on_lds
  PTTS_DAC_LAST_UP_LDS=1 timeout 120 tools/cabi_probe dac $B tag=last_up_on_lds
done
} > gpurun_out/r05_probes13.txt 2>&1
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_fold -o fold -- "$GRAFT_REPO_ROOT"/tools/cabi_probe lm 1 steps=4 > /dev/null 2>&1; grep -h "xfold" /tmp/prof_fold/*/*kernel_stats.csv /tmp/prof_fold/*kernel_stats.csv 2>/dev/null | cut -c1-200 ) >> gpurun_out/r05_probes13.txt 2>&1
grep -E "cabi_probe|xfold" gpurun_out/r05_probes13.txt | cut -c1-250
( timeout 600 python -m pytest tests/test_dac_stage_parity_gpu.py -m gpu -x -q 2>&1 | tail -4; timeout 600 python -m pytest tests/test_lm_gpu.py -m gpu -x -q -k "fused_cross_block_node_single_utterance or single_utterance_gemv_step or mini_width_two_layers or full_mini_v1" 2>&1 | tail -3 ) > gpurun_out/r05_gputest13.txt
cat gpurun_out/r05_gputest13.txt
