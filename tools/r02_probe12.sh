#!/bin/bash
# PMC passes of the bs=1 eager decode step on the final build (refreshes profiles/r02_pmc_step_bs1.json on the box), then the bench line
O=gpurun_out/r02w; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  PROF_B=1 PROF_STEPS=24 timeout 120 rocprofv3 --pmc $C -d $GRAFT_REPO_ROOT/$O/pmc_$C -o p -- python $GRAFT_REPO_ROOT/tools/prof_eager.py > $GRAFT_REPO_ROOT/$O/pmc_$C.log 2>&1
done
cd $GRAFT_REPO_ROOT
F=$(find $O/pmc_FETCH_SIZE -name "*.db" 2>/dev/null | head -1); W=$(find $O/pmc_WRITE_SIZE -name "*.db" 2>/dev/null | head -1)
if [ -n "$F" ] && [ -n "$W" ]; then python tools/pmc_report2.py $F $W 16 57 1 $O/r02_pmc_step_bs1.json > $O/r02_pmc_step_bs1.txt 2>&1 && cp $O/r02_pmc_step_bs1.json profiles/r02_pmc_step_bs1.json; fi
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
tail -3 $O/r02_pmc_step_bs1.txt
timeout 200 python bench.py --steps 3 --warmup 1 > $O/bench.log 2>&1
tail -1 $O/bench.log | cut -c1-400
