#!/bin/bash
# round-2 GPU probe 1: GEMV decode step correctness (existing -m gpu suite) + step latency A/B + per-kernel table
mkdir -p gpurun_out/r02a
O=gpurun_out/r02a
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
for B in 1 8 32; do timeout 300 python tools/step_probe.py $B gemv >> $O/steps.log 2>&1; done
PTTS_NO_GEMV=1 timeout 300 python tools/step_probe.py 1 mfma >> $O/steps.log 2>&1
for S in 2 8; do PTTS_ATTN_SPLITS=$S timeout 300 python tools/step_probe.py 1 gemv_S$S >> $O/steps.log 2>&1; done
PTTS_ATTN_WAVES=8 timeout 300 python tools/step_probe.py 1 gemv_W8 >> $O/steps.log 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof1 -o p -- python $GRAFT_REPO_ROOT/tools/prof_step.py > $GRAFT_REPO_ROOT/$O/prof1.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof1 -name "*.db" | head -1)
python tools/prof_report.py $DB 24 100 > $O/prof1_report.txt 2>&1
rm -rf $O/prof1
tail -5 $O/pytest.log; cat $O/steps.log; head -14 $O/prof1_report.txt
