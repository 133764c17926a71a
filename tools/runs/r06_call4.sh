set -x
cd /root/repo
hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=14 -I include -I parler_tts_amd/csrc tools/gemm_probe.hip -o tools/gemm_probe 2>&1 | tail -5
timeout 600 tools/gemm_probe 40 > gpurun_out/r06_gemm_probe_call4.txt 2>&1
tail -150 gpurun_out/r06_gemm_probe_call4.txt
