cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_glds_preload_ab.txt
: > $O
N=tools/cabi_probe; P=tools/variants/prev52/cabi_probe_prev52
for rep in 1 2 3; do
for B in 32 64; do
  $N t5 $B tag=glds_preload 2>&1 | grep cabi_probe | cut -c1-200 >> $O
  $P t5 $B tag=previous 2>&1 | grep cabi_probe | cut -c1-200 >> $O
  $N lm $B tag=glds_preload 2>&1 | grep cabi_probe | cut -c1-220 >> $O
  $P lm $B tag=previous 2>&1 | grep cabi_probe | cut -c1-220 >> $O
done
done
$N lm 32 large tag=glds_preload 2>&1 | grep cabi_probe | cut -c1-220 >> $O
$P lm 32 large tag=previous 2>&1 | grep cabi_probe | cut -c1-220 >> $O
$N t5 128 tag=glds_preload 2>&1 | grep cabi_probe | cut -c1-200 >> $O
$P t5 128 tag=previous 2>&1 | grep cabi_probe | cut -c1-200 >> $O
$N lm 32 dump=/tmp/a.bin > /dev/null 2>&1; $P lm 32 dump=/tmp/b.bin > /dev/null 2>&1; $N cmp /tmp/a.bin /tmp/b.bin >> $O 2>&1
cat $O
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 900 python -m pytest tests/test_t5_gpu.py tests/test_lm_gpu.py tests/test_generate_gpu.py tests/test_max_context_gpu.py -x -q 2>&1 | grep -v "$F" | tail -5
