#!/bin/bash
# PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs) of the bs=32 eager decode step -> profiles/r02_pmc_step_bs32.json
O=gpurun_out/r02t; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  PROF_B=32 PROF_STEPS=24 timeout 300 rocprofv3 --pmc $C -d $GRAFT_REPO_ROOT/$O/pmc_$C -o p -- python $GRAFT_REPO_ROOT/tools/prof_eager.py > $GRAFT_REPO_ROOT/$O/pmc_$C.log 2>&1
done
cd $GRAFT_REPO_ROOT
F=$(find $O/pmc_FETCH_SIZE -name "*.db" 2>/dev/null | head -1); W=$(find $O/pmc_WRITE_SIZE -name "*.db" 2>/dev/null | head -1)
if [ -n "$F" ] && [ -n "$W" ]; then python tools/pmc_report2.py $F $W 16 57 32 $O/r02_pmc_step_bs32.json > $O/r02_pmc_step_bs32.txt 2>&1; else echo "pmc passes produced no database" > $O/r02_pmc_step_bs32.txt; tail -5 $O/pmc_FETCH_SIZE.log >> $O/r02_pmc_step_bs32.txt; fi
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
cat $O/r02_pmc_step_bs32.txt
