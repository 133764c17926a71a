set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_attn_lds_ab.txt
: > $O
N=tools/cabi_probe; P=tools/variants/prev46/cabi_probe_prev46
for rep in 1 2 3; do
for B in 32 64; do
  $N t5 $B tag=lds_prefetch 2>&1 | grep cabi_probe | cut -c1-200 >> $O
  $P t5 $B tag=previous 2>&1 | grep cabi_probe | cut -c1-200 >> $O
  $N lm $B tag=lds_prefetch 2>&1 | grep cabi_probe | cut -c1-220 >> $O
  $P lm $B tag=previous 2>&1 | grep cabi_probe | cut -c1-220 >> $O
done
done
$N lm 1 tag=lds_prefetch 2>&1 | grep cabi_probe | cut -c1-220 >> $O
$P lm 1 tag=previous 2>&1 | grep cabi_probe | cut -c1-220 >> $O
$N t5 1 tag=lds_prefetch 2>&1 | grep cabi_probe | cut -c1-220 >> $O
$P t5 1 tag=previous 2>&1 | grep cabi_probe | cut -c1-220 >> $O
$N lm 32 large tag=lds_prefetch 2>&1 | grep cabi_probe | cut -c1-220 >> $O
$P lm 32 large tag=previous 2>&1 | grep cabi_probe | cut -c1-220 >> $O
# bit identity: dumps of the prefill logits + 6 eager steps + graph ids
$N lm 32 dump=/tmp/a.bin > /dev/null 2>&1; $P lm 32 dump=/tmp/b.bin > /dev/null 2>&1; $N cmp /tmp/a.bin /tmp/b.bin >> $O 2>&1
$N lm 32 fp32 dump=/tmp/a.bin > /dev/null 2>&1; $P lm 32 fp32 dump=/tmp/b.bin > /dev/null 2>&1; $N cmp /tmp/a.bin /tmp/b.bin >> $O 2>&1
cat $O
timeout 900 python -m pytest tests/test_t5_gpu.py tests/test_lm_gpu.py tests/test_generate_gpu.py -x -q 2>&1 | tail -5
