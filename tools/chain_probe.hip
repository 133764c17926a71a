// Standalone probe (not part of the product): the decode GEMM kernel inside a realistic DEPENDENT chain - 170 graph nodes,
// each reads the vector the previous node wrote (cold: other XCDs wrote it) and streams its own never-before-touched
// weights (rotating over 680 MB: HBM-cold, MALL-cold). Prints per-node wall time and the s_memtime phase stamps of
// workgroup 0 / thread 0 of one node in the middle of the chain.
// Build: hipcc --offload-arch=gfx950 -O3 -w -std=c++17 -o tools/chain_probe tools/chain_probe.hip
#define PTTS_TIMING 1
#include "../parler_tts_amd/csrc/ptts_lm_kernels.h"
#include <stdio.h>
thread_local std::string g_ptts_err;
int ptts_fail(int code, const char*, ...) { return code; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int PRO, int EPI> static void launch(const GemmArgs& a, int Wv, size_t sh, hipStream_t st) {
  auto k = gemm_strip_kernel_bv<bf16_t, PRO, EPI, 1, true>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(k, dim3(a.N / 16), dim3(Wv * 64), sh, st, a);
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const int NODES = 170;
  const size_t slot = (size_t)4096 * 1024 * 2;  // 8 MB: the largest decode matrix
  char* W; float *xa, *xb, *h, *gamma, *beta; long long* dbg;
  CK(hipMalloc(&W, slot * 85)); CK(hipMemset(W, 0, slot * 85));
  CK(hipMalloc(&xa, 4096 * 4 * 4)); CK(hipMalloc(&xb, 4096 * 4 * 4)); CK(hipMalloc(&h, 4096 * 4 * 4));
  CK(hipMalloc(&gamma, 4096 * 4)); CK(hipMalloc(&beta, 4096 * 4)); CK(hipMalloc(&dbg, 64 * 8));
  CK(hipMemset(xa, 0, 4096 * 16)); CK(hipMemset(xb, 0, 4096 * 16)); CK(hipMemset(h, 0, 4096 * 16)); CK(hipMemset(gamma, 0, 4096 * 4)); CK(hipMemset(beta, 0, 4096 * 4));
  struct Case { const char* name; int N, K, pro, epi; } cases[] = {
      {"LN   +STORE N=1024 K=1024 ( 2 MB)", 1024, 1024, PRO_LN, EPI_STORE},   {"PLAIN+STORE N=1024 K=1024 ( 2 MB)", 1024, 1024, PRO_PLAIN, EPI_STORE},
      {"PLAIN+RESID N=1024 K=1024 ( 2 MB)", 1024, 1024, PRO_PLAIN, EPI_RESID}, {"LN   +STORE N=3072 K=1024 ( 6 MB)", 3072, 1024, PRO_LN, EPI_STORE},
      {"LN   +GELU  N=4096 K=1024 ( 8 MB)", 4096, 1024, PRO_LN, EPI_GELU},    {"PLAIN+RESID N=1024 K=4096 ( 8 MB)", 1024, 4096, PRO_PLAIN, EPI_RESID}};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (auto& cs : cases) {
    const int nfrag = cs.K / 32;
    const int wmax = (cs.pro == PRO_PLAIN) ? 16 : 8;
    int Wv = 0;
    for (int w = wmax; w >= 2; --w) if (nfrag % (8 * w) == 0) { Wv = w; break; }
    const size_t row_bytes = (size_t)cs.K * 2 + 16, sh = row_bytes + (size_t)Wv * 1024;
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < NODES; ++i) {
      GemmArgs a = {}; a.W = W + (size_t)(i % 85) * slot; a.x = (i & 1) ? xb : xa; a.x_ld = cs.K; a.x_row_mul = 1; a.gamma = gamma; a.beta = beta;
      a.out = cs.epi == EPI_RESID ? h : ((i & 1) ? xa : xb); a.out_ld = cs.N; a.M = 1; a.N = cs.N; a.K = cs.K; a.invK = 1.0f / cs.K;
      a.rows_per_pass = 1; a.frags_per_wave = nfrag / Wv; a.dbg = (i == 100) ? dbg : nullptr;
      if (cs.pro == PRO_LN && cs.epi == EPI_STORE) launch<PRO_LN, EPI_STORE>(a, Wv, sh, st);
      else if (cs.pro == PRO_LN) launch<PRO_LN, EPI_GELU>(a, Wv, sh, st);
      else if (cs.epi == EPI_STORE) launch<PRO_PLAIN, EPI_STORE>(a, Wv, sh, st);
      else launch<PRO_PLAIN, EPI_RESID>(a, Wv, sh, st);
    }
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    for (int i = 0; i < 10; ++i) hipGraphLaunch(ex, st);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < 100; ++i) hipGraphLaunch(ex, st);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long t[16]; hipMemcpy(t, dbg, 128, hipMemcpyDeviceToHost);
    printf("%s W=%2d: %.2f us per node | ticks: issue-W %lld, stage %lld, barrier %lld, mfma(+W wait) %lld, reduce+store %lld, total %lld\n", cs.name, Wv,
           ms * 1e3f / 100 / NODES, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[5] - t[0]);
    if (cs.pro == PRO_LN) printf("      LN stage: loads issued +%lld, rendezvous +%lld, rows arrived+accumulated +%lld, wave reductions +%lld, normalise +%lld, weight issue + LDS stores +%lld\n",
           t[6] - t[1], t[7] - t[6], t[8] - t[7], t[9] - t[8], t[10] - t[9], t[2] - t[10]);
    hipGraphExecDestroy(ex); hipGraphDestroy(g);
  }
  return 0;
}
