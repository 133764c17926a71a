#!/bin/bash
# round 5, GPU call 8 (records of the final tree): T5 norm fold re-measured, rocprofv3 summary of the bench command, kernel table of the
# time-to-first-token path (T5 + prefill), step table at 128 utterances, whole GPU suite, smoke(), the whole bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
{
timeout 300 python tools/ttft_probe5.py t5_fold 1
PTTS_T5_NO_FOLD=1 timeout 300 python tools/ttft_probe5.py t5_rows_prep 1
} > gpurun_out/r05_probes8.txt 2>&1
grep ttft_probe5 gpurun_out/r05_probes8.txt | cut -c1-200
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/pb; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pb -o p -- python $R/bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > /tmp/pb.log 2>&1
python $R/tools/prof_report.py $(find /tmp/pb -name '*.db' | head -1) 30 1734 > $R/gpurun_out/r05_bench_bs1_rocprof_summary.txt 2>&1
rm -rf /tmp/pp; PROF_B=1 PROF_N=20 timeout 400 rocprofv3 --kernel-trace -d /tmp/pp -o p -- python $R/tools/prof_prefill.py > /tmp/pp.log 2>&1
python $R/tools/prof_report.py $(find /tmp/pp -name '*.db' | head -1) 30 > $R/gpurun_out/r05_prefill_kernels_bs1.txt 2>&1
rm -rf /tmp/pp32; PROF_B=32 PROF_N=8 timeout 400 rocprofv3 --kernel-trace -d /tmp/pp32 -o p -- python $R/tools/prof_prefill.py > /tmp/pp32.log 2>&1
python $R/tools/prof_report.py $(find /tmp/pp32 -name '*.db' | head -1) 30 > $R/gpurun_out/r05_prefill_kernels_bs32.txt 2>&1
rm -rf /tmp/pl128; PROF_B=128 PROF_STEPS=20 timeout 400 rocprofv3 --kernel-trace -d /tmp/pl128 -o p -- python $R/tools/prof_step.py > /dev/null 2>&1
python $R/tools/prof_report.py $(find /tmp/pl128 -name '*.db' | head -1) 14 420 > $R/gpurun_out/r05_step_bf16_bs128.txt 2>&1
cd $R
head -14 gpurun_out/r05_bench_bs1_rocprof_summary.txt | cut -c1-150; tail -2 gpurun_out/r05_bench_bs1_rocprof_summary.txt
head -22 gpurun_out/r05_prefill_kernels_bs1.txt | cut -c1-150
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest8.txt 2>&1; echo "suite rc=$?" >> gpurun_out/r05_gputest8.txt
tail -6 gpurun_out/r05_gputest8.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke8.txt 2>&1; tail -2 gpurun_out/r05_smoke8.txt
( time timeout 1500 python bench.py ) > gpurun_out/r05_bench8.json.log 2> gpurun_out/r05_bench8.err
tail -3 gpurun_out/r05_bench8.err; tail -c 1500 gpurun_out/r05_bench8.json.log
