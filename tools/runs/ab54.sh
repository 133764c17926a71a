cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_attn_prep_preload_ab.txt
: > $O
N=tools/cabi_probe; P=tools/variants/prev54/cabi_probe_prev54
for rep in 1 2 3; do
for B in 1 32; do
  $N t5 $B tag=preload 2>&1 | grep cabi_probe | cut -c1-200 >> $O
  $P t5 $B tag=previous 2>&1 | grep cabi_probe | cut -c1-200 >> $O
  $N lm $B tag=preload 2>&1 | grep cabi_probe | cut -c1-220 >> $O
  $P lm $B tag=previous 2>&1 | grep cabi_probe | cut -c1-220 >> $O
done
done
for B in 4 64; do
  $N lm $B tag=preload 2>&1 | grep cabi_probe | cut -c1-220 >> $O
  $P lm $B tag=previous 2>&1 | grep cabi_probe | cut -c1-220 >> $O
done
$N lm 32 large tag=preload 2>&1 | grep cabi_probe | cut -c1-220 >> $O
$P lm 32 large tag=previous 2>&1 | grep cabi_probe | cut -c1-220 >> $O
for B in 1 32; do
$N lm $B dump=/tmp/a.bin > /dev/null 2>&1; $P lm $B dump=/tmp/b.bin > /dev/null 2>&1; $N cmp /tmp/a.bin /tmp/b.bin >> $O 2>&1
$N lm $B fp32 dump=/tmp/a.bin > /dev/null 2>&1; $P lm $B fp32 dump=/tmp/b.bin > /dev/null 2>&1; $N cmp /tmp/a.bin /tmp/b.bin >> $O 2>&1
done
tools/attn_probe >> $O 2>&1
cat $O
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 1200 python -m pytest tests/test_t5_gpu.py tests/test_lm_gpu.py tests/test_generate_gpu.py tests/test_max_context_gpu.py -x -q 2>&1 | grep -v "$F" | tail -5
