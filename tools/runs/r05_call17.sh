#!/bin/bash
# round 5, GPU call 17 (records of the tree after kernarg preload + DAC epilogues): the whole bench line, GPU suite, smoke(), rocprofv3 summary of the bench
# command, kernel table of the time-to-first-token path (incl. the tiled fold kernels), DAC kernel tables (batch 32 / 1) + MFMA-busy counters per DAC kernel
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( time timeout 1500 python bench.py ) > gpurun_out/r05_bench17.json.log 2> gpurun_out/r05_bench17.err
tail -3 gpurun_out/r05_bench17.err; head -c 1800 gpurun_out/r05_bench17.json.log; echo
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest17.txt 2>&1; echo "suite rc=$?" >> gpurun_out/r05_gputest17.txt
tail -6 gpurun_out/r05_gputest17.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke17.txt 2>&1; tail -2 gpurun_out/r05_smoke17.txt
cd /tmp
rm -rf /tmp/pb; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pb -o p -- python $R/bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > /tmp/pb.log 2>&1
python $R/tools/prof_report.py $(find /tmp/pb -name '*.db' | head -1) 30 1734 > $R/gpurun_out/r05_bench_bs1_rocprof_summary_v2.txt 2>&1
rm -rf /tmp/pp; PROF_B=1 PROF_N=20 timeout 400 rocprofv3 --kernel-trace -d /tmp/pp -o p -- python $R/tools/prof_prefill.py > /tmp/pp.log 2>&1
python $R/tools/prof_report.py $(find /tmp/pp -name '*.db' | head -1) 30 > $R/gpurun_out/r05_prefill_kernels_bs1_v2.txt 2>&1
rm -rf /tmp/pd32; timeout 300 rocprofv3 --kernel-trace -d /tmp/pd32 -o p -- $R/tools/cabi_probe dac 32 reps=3 > /dev/null 2>&1
python $R/tools/prof_report.py $(find /tmp/pd32 -name '*.db' | head -1) 14 > $R/gpurun_out/r05_dac_kernels_bs32.txt 2>&1
rm -rf /tmp/pd1; timeout 300 rocprofv3 --kernel-trace -d /tmp/pd1 -o p -- $R/tools/cabi_probe dac 1 reps=10 > /dev/null 2>&1
python $R/tools/prof_report.py $(find /tmp/pd1 -name '*.db' | head -1) 14 > $R/gpurun_out/r05_dac_kernels_bs1.txt 2>&1
rm -rf /tmp/pm32; timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_VALU --kernel-trace -d /tmp/pm32 -o p -- $R/tools/cabi_probe dac 32 reps=2 > /tmp/pm32.log 2>&1
python $R/tools/pmc_mfma_report.py $(find /tmp/pm32 -name "*.db" | head -1) > $R/gpurun_out/r05_pmc_dac_mfma.txt 2>&1
cd $R
head -16 gpurun_out/r05_bench_bs1_rocprof_summary_v2.txt | cut -c1-150; tail -2 gpurun_out/r05_bench_bs1_rocprof_summary_v2.txt
grep -i "xfold" gpurun_out/r05_prefill_kernels_bs1_v2.txt | cut -c1-150
head -16 gpurun_out/r05_dac_kernels_bs32.txt | cut -c1-150
head -16 gpurun_out/r05_pmc_dac_mfma.txt | cut -c1-170
