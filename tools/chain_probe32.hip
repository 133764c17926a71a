// Standalone probe (not part of the product): the batch-32 decode GEMM (gemm_strip_kernel, MTP = 2, prepared bf16 rows = PRO_COPY, and the
// producer-statistics LayerNorm prologue PRO_LNS) inside a cold DEPENDENT chain of 219 graph nodes - what tools/chain_probe.hip does for
// batch 1. Every node reads the 32-row block the previous node wrote (other XCDs wrote it) and streams its own weights (rotating over
// 680 MB: HBM-cold, MALL-cold). Prints per-node wall time and the s_memtime phase stamps of workgroup 0 / thread 0 of node 100:
//   issue-W (residual prefetch + weight issue), stage (LNS prologue -> LDS; 0 for PRO_COPY), barrier, mfma (B fragments from L2 + weight wait
//   + MFMAs), reduce+store (cross-wave LDS reduction, epilogue, strip statistics).
// Purpose: decide what to cut in the 5.3-7.4 us batch-32 nodes (profiles/r02_step_bf16_bs32_lns_v1.txt) before rewriting them.
// Build: hipcc --offload-arch=gfx950 -O3 -w -std=c++17 -o tools/chain_probe32 tools/chain_probe32.hip
#define PTTS_TIMING 1
#include "../parler_tts_amd/csrc/ptts_lm_kernels.h"
#include <stdio.h>
thread_local std::string g_ptts_err;
int ptts_fail(int code, const char*, ...) { return code; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int PRO, int EPI> static void launch(const GemmArgs& a, int Wv, size_t sh, dim3 grid, hipStream_t st) {
  auto k = gemm_strip_kernel_bv<bf16_t, PRO, EPI, 2, true>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(k, grid, dim3(Wv * 64), sh, st, a);
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const int NODES = 219, M = 32;
  const size_t slot = (size_t)4096 * 1024 * 2;  // 8 MB: the largest decode matrix
  char* W; char *xa, *xb; float *h, *gamma, *beta, *lnstat, *part; long long* dbg;
  const size_t act = (size_t)M * 4096 * 4;  // one activation block: fp32 [32][4096] covers every in / out shape below
  CK(hipMalloc(&W, slot * 85)); CK(hipMemset(W, 0, slot * 85));
  CK(hipMalloc(&xa, act)); CK(hipMalloc(&xb, act)); CK(hipMalloc(&h, act)); CK(hipMalloc(&part, act * 4));
  CK(hipMalloc(&gamma, 4096 * 4)); CK(hipMalloc(&beta, 4096 * 4)); CK(hipMalloc(&lnstat, (size_t)M * 256 * 2 * 4)); CK(hipMalloc(&dbg, 64 * 8));
  CK(hipMemset(xa, 0, act)); CK(hipMemset(xb, 0, act)); CK(hipMemset(h, 0, act)); CK(hipMemset(part, 0, act * 4));
  CK(hipMemset(gamma, 0, 4096 * 4)); CK(hipMemset(beta, 0, 4096 * 4)); CK(hipMemset(lnstat, 0, (size_t)M * 256 * 2 * 4)); CK(hipMemset(dbg, 0, 64 * 8));
  struct Case { const char* name; int N, K, pro, epi, ksplit; } cases[] = {
      {"COPY +STORE      N=3072 K=1024 ( 6 MB) QKV       ", 3072, 1024, PRO_COPY, EPI_STORE, 0},
      {"COPY +RESID+stat N=1024 K=1024 ( 2 MB) out_proj  ", 1024, 1024, PRO_COPY, EPI_RESID, 0},
      {"LNS  +STORE      N=1024 K=1024 ( 2 MB) cross q   ", 1024, 1024, PRO_LNS, EPI_STORE, 0},
      {"LNS  +GELU(bf16) N=4096 K=1024 ( 8 MB) fc1       ", 4096, 1024, PRO_LNS, EPI_GELU_WT, 0},
      {"COPY +STORE x4   N=1024 K=4096 ( 8 MB) fc2 splitK", 1024, 4096, PRO_COPY, EPI_STORE, 4}};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (auto& cs : cases) {
    const int nfrag_all = cs.K / 32, nfrag = cs.ksplit ? nfrag_all / cs.ksplit : nfrag_all;  // fragments one workgroup covers
    int Wv = 0;
    for (int w = 8; w >= 2; --w) if (nfrag % (8 * w) == 0) { Wv = w; break; }
    if (!Wv) { printf("%s: no FULL wave split\n", cs.name); continue; }
    const size_t row_bytes = cs.pro == PRO_COPY ? 0 : (size_t)cs.K * 2 + 16;
    const size_t sh = (size_t)M * row_bytes + (size_t)Wv * 2 * 1024;
    const dim3 grid(cs.N / 16, cs.ksplit ? cs.ksplit : 1);
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < NODES; ++i) {
      GemmArgs a = {};
      a.W = W + (size_t)(i % 85) * slot; a.x = reinterpret_cast<const float*>((i & 1) ? xb : xa); a.x_ld = cs.K; a.x_row_mul = 1; a.gamma = gamma; a.beta = beta;
      a.lnstat = cs.pro == PRO_LNS ? lnstat : nullptr;
      a.out = cs.epi == EPI_RESID ? h : (cs.ksplit ? part : reinterpret_cast<float*>((i & 1) ? xa : xb));
      a.out_ld = cs.N; a.M = M; a.N = cs.N; a.K = cs.K; a.invK = 1.0f / cs.K; a.rows_per_pass = M; a.frags_per_wave = nfrag / Wv;
      a.stats_out = cs.epi == EPI_RESID ? lnstat : nullptr;
      if (cs.ksplit) { a.ksplit = nfrag; a.out_split_stride = (long long)M * cs.N; }
      a.dbg = (i == 100) ? dbg : nullptr;
      if (cs.pro == PRO_COPY && cs.epi == EPI_STORE) launch<PRO_COPY, EPI_STORE>(a, Wv, sh, grid, st);
      else if (cs.pro == PRO_COPY) launch<PRO_COPY, EPI_RESID>(a, Wv, sh, grid, st);
      else if (cs.epi == EPI_STORE) launch<PRO_LNS, EPI_STORE>(a, Wv, sh, grid, st);
      else launch<PRO_LNS, EPI_GELU_WT>(a, Wv, sh, grid, st);
    }
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    for (int i = 0; i < 10; ++i) hipGraphLaunch(ex, st);
    CK(hipStreamSynchronize(st));
    hipEventRecord(e0, st);
    for (int i = 0; i < 100; ++i) hipGraphLaunch(ex, st);
    hipEventRecord(e1, st); CK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long t[16]; CK(hipMemcpy(t, dbg, 128, hipMemcpyDeviceToHost));
    const bool copy = cs.pro == PRO_COPY;  // no staging phase: stamp 2 is not written
    printf("%s W=%d grid %dx%d: %.2f us per node | ticks: issue-W %lld, stage %lld, barrier %lld, mfma(+B from L2, W wait) %lld, reduce+store %lld, total %lld\n",
           cs.name, Wv, grid.x, grid.y, ms * 1e3f / 100 / NODES, t[1] - t[0], copy ? 0LL : t[2] - t[1], copy ? t[3] - t[1] : t[3] - t[2], t[4] - t[3], t[5] - t[4],
           t[5] - t[0]);
    hipGraphExecDestroy(ex); hipGraphDestroy(g);
  }
  return 0;
}
